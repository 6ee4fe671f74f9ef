#!/bin/bash
# compact (live-only) appearance sort: backward parity tests on the product library, then A/B RDRF_SORT_COMPACT=0/1 (tools build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_deterministic.py -q -m gpu 2>&1 | tail -4 > gpurun_out/sortc_tests.txt
cat gpurun_out/sortc_tests.txt
export RDRF_LIB=$PWD/robust-dynrf_amd/librodynrf_tools.so
{
for stage in stage0 final; do for i in 1 2; do for x in 0 1; do
  RDRF_SORT_COMPACT=$x timeout 300 python bench.py --stage $stage --steps 20 --warmup 5 --no-cpu-baseline --no-final-stage --no-render --no-sparse --no-liveness-leg >/dev/null 2>&1
  python - "$stage RDRF_SORT_COMPACT=$x" <<'PY'
import json, sys
d = json.load(open("bench_detail.json")); r = d["roofline"]["kernel_ms_per_step"]
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in r.items() if k in ("scatter_dyn_app", "sort", "scatter_dyn_density")})
PY
done; done; done
} > gpurun_out/sortc_ab.txt 2>&1
cat gpurun_out/sortc_ab.txt
