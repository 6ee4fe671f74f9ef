#!/bin/bash
# render + training A/B of two builds on one box: tools/ab_flat.sh [lib ...] (default: librodynrf_base.so = HEAD~, librodynrf.so)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
LIBS=${@:-"robust-dynrf_amd/librodynrf_base.so robust-dynrf_amd/librodynrf.so"}
for i in 1 2; do
  for lib in $LIBS; do
    echo "== $lib (run $i)"
    RDRF_LIB=$PWD/$lib timeout 300 python tools/render_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids"
  done
done
for i in 1 2; do tools/ab_lib.sh $LIBS; done
