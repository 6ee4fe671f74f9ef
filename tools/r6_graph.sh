#!/bin/bash
# Trainer(graph=True): tests, then ms/iteration of the launch-bound stages eager vs captured
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_losses.py -x -q 2>&1 | tail -25 > gpurun_out/graph_tests.txt
cat gpurun_out/graph_tests.txt
: > gpurun_out/graph_ab.txt
for cfg in "nvidia_no_poses stage0" "davis stage0" "nvidia stage0"; do
  set -- $cfg
  for mode in "" "--graph"; do
    timeout 400 python bench.py --full-line --config $1 --stage $2 --steps 40 --warmup 6 --no-cpu-baseline --no-final-stage --no-render --no-sparse --no-roofline $mode 2>&1 | tail -1 > gpurun_out/cfgt.log
    python - "$1" "$2" "$mode" <<'PY' | tee -a gpurun_out/graph_ab.txt
import json, sys
try:
    d = json.loads(open("gpurun_out/cfgt.log").read().strip().splitlines()[-1])
    c = d["config"]
    print(sys.argv[1], sys.argv[2], sys.argv[3] or "eager", "grid", c.get("grid"), "S", c.get("samples_per_ray"), "rays", c.get("global_batch"), "ms/step", round(d["ms_per_step"], 3),
          "rays/s", round(d["value"]), "liveness_exploited ms", round(d.get("liveness_exploited", {}).get("ms_per_step", 0), 3), "loss", c.get("final_loss"))
except Exception as e:
    print(sys.argv[1:], "ERR", e, open("gpurun_out/cfgt.log").read()[-1500:])
PY
  done
done
