#!/bin/bash
# S = 13 stages: scatter modes under eager / captured iterations
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ulimit -c 0
mkdir -p gpurun_out
: > gpurun_out/graph_scatter_ab.txt
for cfg in nvidia_no_poses davis; do
  for mode in "--scatter auto" "--scatter auto --graph" "--scatter sorted --graph" "--scatter sorted_plain --graph" "--scatter sorted"; do
    timeout 300 python bench.py --full-line --config $cfg --stage stage0 --steps 60 --warmup 8 --no-cpu-baseline --no-final-stage --no-render --no-sparse --no-roofline --no-liveness-leg $mode 2>&1 | tail -1 > gpurun_out/cfgt.log
    python - "$cfg" "$mode" <<'PY' | tee -a gpurun_out/graph_scatter_ab.txt
import json, sys
try:
    d = json.loads(open("gpurun_out/cfgt.log").read().strip().splitlines()[-1])
    print(sys.argv[1], sys.argv[2], "ms/step", round(d["ms_per_step"], 3), "rays/s", round(d["value"]), "loss", round(d["config"]["final_loss"], 4))
except Exception as e:
    print(sys.argv[1:], "ERR", e, open("gpurun_out/cfgt.log").read()[-800:])
PY
  done
done
