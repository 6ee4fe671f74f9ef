#!/bin/bash
# A/B (tools build) of distinct wave priorities (RDRF_WPRIO) x tile queue (RDRF_DYNQ) x tile size (RDRF_SA16) on the static
# appearance kernel -> gpurun_out/wprio_ab.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export RDRF_LIB=$PWD/robust-dynrf_amd/librodynrf_tools.so
{
for shape in "16384 115 141,157,94" "16384 270 331,368,220"; do
for i in 1 2; do for x in 0 1; do for cfg in "0 0" "1 0" "1 1" "0 1"; do
  set -- $cfg
  echo "== RDRF_SA16=$x RDRF_DYNQ=$1 RDRF_WPRIO=$2 [$shape]"; RDRF_SA16=$x RDRF_DYNQ=$1 RDRF_WPRIO=$2 timeout 300 python tools/fwd_ab.py $shape 2>&1 | grep -E "train|infer" | tail -2 | sed 's/| time_branch.*//'
done; done; done; done
} > gpurun_out/wprio_ab.txt 2>&1
cat gpurun_out/wprio_ab.txt
