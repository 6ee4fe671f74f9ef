#!/bin/bash
# ms/iteration and rays/s of the other shipped configs: tools/configs_table.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
for cfg in "nvidia_no_poses stage0" "nvidia_no_poses stage0 --graph" "nvidia_no_poses final" "davis stage0" "davis stage0 --graph" "davis final"; do
  set -- $cfg
  timeout 400 python bench.py --full-line --config $1 --stage $2 $3 --steps 20 --warmup 3 --no-cpu-baseline --no-final-stage --no-render --no-sparse --no-roofline 2>&1 | tail -1 > gpurun_out/cfgt.log
  python - "$1" "$2$3" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/cfgt.log").read().strip().splitlines()[-1])
    c = d["config"]
    print(sys.argv[1], sys.argv[2], "grid", c.get("grid"), "S", c.get("samples_per_ray"), "rays", c.get("global_batch"), "ms/step", round(d["ms_per_step"], 2), "rays/s", round(d["value"]),
          "liveness_exploited", round(d.get("liveness_exploited_value", 0)))
except Exception as e:
    print(sys.argv[1:], "ERR", e, open("gpurun_out/cfgt.log").read()[-600:])
PY
done
