#!/bin/bash
# interleaved A/B of the static appearance kernel (tools build): RDRF_SA16 (0: 32-sample tiles, 1: 16-sample tiles) x
# RDRF_DYNQ (0: static tile stride, 1: per-workgroup tile queue), stage-0 and final-stage shapes -> gpurun_out/sa16_ab.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export RDRF_LIB=$PWD/robust-dynrf_amd/librodynrf_tools.so
{
for shape in "16384 115 141,157,94" "16384 270 331,368,220"; do
for i in 1 2; do for x in 0 1; do for q in 0 1; do
  echo "== RDRF_SA16=$x RDRF_DYNQ=$q  [$shape]"; RDRF_SA16=$x RDRF_DYNQ=$q timeout 300 python tools/fwd_ab.py $shape 2>&1 | grep -E "train|infer" | tail -2 | sed 's/| time_branch.*//'
done; done; done; done
} > gpurun_out/sa16_ab.txt 2>&1
cat gpurun_out/sa16_ab.txt
