#!/bin/bash
# round 6: k_dw3 (half-stage LDS-DMA dW kernel): gradient parity tests on the product library (k_dw3 is its default), then an
# interleaved A/B of the training step against k_dw2 (tools build, RDRF_DW3=0/1), stage 0 and final stage
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_features.py tests/test_gpu_flow.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/dw3_tests.txt
cat gpurun_out/dw3_tests.txt
export RDRF_LIB=$PWD/robust-dynrf_amd/librodynrf_tools.so
{
for stage in stage0 final; do for i in 1 2; do for x in 0 1; do
  RDRF_DW3=$x timeout 300 python bench.py --stage $stage --steps 20 --warmup 5 --no-cpu-baseline --no-final-stage --no-render --no-sparse >/dev/null 2>&1
  python - "$stage RDRF_DW3=$x" <<'PY'
import json, sys
d = json.load(open("bench_detail.json")); r = d["roofline"]["kernel_ms_per_step"]
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in r.items() if k.startswith("dw_")}, "dw sum", round(sum(v for k, v in r.items() if k.startswith("dw_")), 3))
PY
done; done; done
} > gpurun_out/dw3_ab.txt 2>&1
cat gpurun_out/dw3_ab.txt
