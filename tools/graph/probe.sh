#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
ulimit -c 0
: > gpurun_out/graph_probe.txt
for w in "$@"; do
  timeout 120 python tools/graph/probe.py $w ${PROBE_CFG:-nvidia_no_poses} 2>&1 | grep -v "^  File\|amdgpu.ids\|Extension modules\|^frame\|UserWarning\|Consider using\|print(what" | grep "^$w\|rror" | tail -4 >> gpurun_out/graph_probe.txt
  echo "rc $w ${PIPESTATUS[0]}" >> gpurun_out/graph_probe.txt
done
cat gpurun_out/graph_probe.txt
