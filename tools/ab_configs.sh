#!/bin/bash
# RDRF_BATCH_PASSES=0 / 1 on the secondary configs, one box: tools/ab_configs.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
for cfg in "nvidia_no_poses stage0" "nvidia_no_poses final" "davis stage0" "davis final"; do
  set -- $cfg
  for bp in 0 1; do
    RDRF_BATCH_PASSES=$bp timeout 300 python bench.py --full-line --config $1 --stage $2 --steps 20 --warmup 3 --no-cpu-baseline --no-final-stage --no-render --no-sparse --no-roofline 2>&1 | tail -1 > gpurun_out/abc.log
    python - "$1" "$2" "$bp" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/abc.log").read().strip().splitlines()[-1])
    print(sys.argv[1], sys.argv[2], "batch_passes", sys.argv[3], "ms/step", round(d["ms_per_step"], 3), "rays/s", round(d["value"]))
except Exception as e:
    print(sys.argv[1:], "ERR", e, open("gpurun_out/abc.log").read()[-600:])
PY
  done
done
