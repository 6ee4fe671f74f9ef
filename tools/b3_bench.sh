#!/bin/bash
# quick A/B record of a library build: stage-0 / final-stage step and the MLP kernels' times (bf16 x 3 work)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ulimit -c 0
mkdir -p gpurun_out
TAG=${1:-b3}
python bench.py --full-line --steps 30 --warmup 5 --no-cpu-baseline --no-render --no-sparse --no-graph-leg --no-liveness-leg 2>&1 | tail -1 > gpurun_out/${TAG}_bench.json
python - gpurun_out/${TAG}_bench.json <<'PY' | tee gpurun_out/${TAG}_bench.txt
import json, sys
d = json.loads(open(sys.argv[1]).read())
print("stage0 ms", round(d["ms_per_step"], 3), "final ms", round(d["final_stage"]["ms_per_step"], 3), "schedule-weighted rays/s", round(d["schedule_weighted"]["value"]))
keys = ("dyn_density", "dyn_app", "static_app", "scene_flow", "dyn_heads_bwd", "dyn_warp_bwd", "dyn_app_bwd", "static_app_bwd", "scene_flow_bwd", "dw_dyn")
for name, r in (("stage0", d["roofline"]), ("final", d["final_stage"]["roofline"])):
    km = r["kernel_ms_per_step"]
    print(name, {k: round(km[k], 3) for k in keys})
PY
