#!/bin/bash
# round 6: k_static_app16 (16-sample tiles, 4 waves/SIMD) -- parity tests on the product library, then an interleaved A/B
# of the forward kernels (tools build: RDRF_SA16=0 -> 32-sample tiles, 1 -> 16-sample tiles) at stage 0 and the final stage
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_backward.py tests/test_gpu_features.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/sa16_tests.txt
cat gpurun_out/sa16_tests.txt
export RDRF_LIB=$PWD/robust-dynrf_amd/librodynrf_tools.so
{
for i in 1 2; do for x in 0 1; do
  echo "== RDRF_SA16=$x stage0"; RDRF_SA16=$x timeout 300 python tools/fwd_ab.py 16384 115 141,157,94 2>&1 | grep -E "train|infer" | sed 's/dyn_density.*//'
done; done
for i in 1 2; do for x in 0 1; do
  echo "== RDRF_SA16=$x final"; RDRF_SA16=$x timeout 300 python tools/fwd_ab.py 16384 270 331,368,220 2>&1 | grep -E "train|infer" | sed 's/dyn_density.*//'
done; done
} > gpurun_out/sa16_ab.txt 2>&1
cat gpurun_out/sa16_ab.txt
