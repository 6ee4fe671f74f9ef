#!/bin/bash
# A/B (tools build): per-workgroup tile / ray queue (RDRF_DYNQ=1, default) vs static stride (0) in every persistent MLP kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export RDRF_LIB=$PWD/robust-dynrf_amd/librodynrf_tools.so
{
for stage in stage0 final; do for i in 1 2 3; do for x in 0 1; do
  RDRF_DYNQ=$x timeout 300 python bench.py --stage $stage --steps 20 --warmup 5 --no-cpu-baseline --no-final-stage --no-render --no-sparse --no-liveness-leg >/dev/null 2>&1
  python - "$stage RDRF_DYNQ=$x" <<'PY'
import json, sys
d = json.load(open("bench_detail.json")); r = d["roofline"]["kernel_ms_per_step"]
keys = ("dyn_density", "dyn_app", "static_app", "dyn_heads_bwd", "dyn_warp_bwd", "dyn_app_bwd", "static_app_bwd", "scene_flow")
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), "sum", round(sum(r[k] for k in keys), 3), {k: round(r[k], 3) for k in keys})
PY
done; done; done
} > gpurun_out/dynq_ab.txt 2>&1
cat gpurun_out/dynq_ab.txt
