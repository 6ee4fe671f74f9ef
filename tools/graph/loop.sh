#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
ulimit -c 0
: > gpurun_out/graph_loop.txt
while [ $# -gt 0 ]; do
  echo "== $1" >> gpurun_out/graph_loop.txt
  timeout 200 python tools/graph/loop.py $1 2>&1 | grep -v "^  File\|amdgpu.ids\|Extension modules" | tail -${TAILN:-8} >> gpurun_out/graph_loop.txt
  shift
done
cat gpurun_out/graph_loop.txt
