#!/bin/bash
# render legs (whole frame, 512-ray chunks) for several builds: tools/ab_render_libs.sh lib1.so lib2.so ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for i in 1 2; do for lib in "$@"; do
  echo "== $lib"
  RDRF_LIB=$PWD/$lib python tools/render_bench.py 2>/dev/null | grep -E "^chunk|native"
done; done
