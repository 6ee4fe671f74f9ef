#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for tpw in 1 2 3 4; do
  echo "RDRF_GEO_TPW=$tpw"
  RDRF_LIB=$PWD/robust-dynrf_amd/librodynrf_tools.so RDRF_GEO_TPW=$tpw python tools/render_bench.py chunk512 2>/dev/null | grep -E "native|chunk    512"
done
