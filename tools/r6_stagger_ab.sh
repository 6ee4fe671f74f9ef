#!/bin/bash
# A/B (tools build) of the start-up stagger of co-resident waves (RDRF_STAGGER x 8128 cycles) on the static appearance kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export RDRF_LIB=$PWD/robust-dynrf_amd/librodynrf_tools.so
{
for shape in "16384 115 141,157,94" "16384 270 331,368,220"; do
for x in 0 1; do for st in 0 1 2 3 5 8 12; do
  echo "== RDRF_SA16=$x RDRF_STAGGER=$st [$shape]"; RDRF_SA16=$x RDRF_STAGGER=$st timeout 300 python tools/fwd_ab.py $shape 2>&1 | grep -E "train|infer" | tail -2 | sed 's/| time_branch.*//'
done; done; done
} > gpurun_out/stagger_ab.txt 2>&1
cat gpurun_out/stagger_ab.txt
