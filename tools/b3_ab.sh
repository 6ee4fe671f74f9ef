#!/bin/bash
# interleaved A/B on one box: heads' first layers on the fp32 pipe (abl_headsf32.so) against bf16 x 3 (abl_b3final.so), stage 0 and final
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ulimit -c 0
mkdir -p gpurun_out
: > gpurun_out/b3_ab.txt
for stage in stage0 final; do
  for r in 1 2 3; do
    for lib in ${AB_LIBS:-abl_headsf32 abl_b3final}; do
      RDRF_LIB=$PWD/robust-dynrf_amd/$lib.so timeout 300 python bench.py --full-line --stage $stage --config nvidia --steps 30 --warmup 6 --no-cpu-baseline --no-final-stage --no-render --no-sparse --no-graph-leg --no-liveness-leg 2>&1 | tail -1 > gpurun_out/ablib.log
      python - "$lib" "$stage" "$r" <<'PY' | tee -a gpurun_out/b3_ab.txt
import json, os, sys
try:
    d = json.loads(open("gpurun_out/ablib.log").read().strip().splitlines()[-1]); r = d["roofline"]["kernel_ms_per_step"]
    keys = os.environ.get("AB_KEYS", "dyn_density dyn_heads_bwd dyn_warp_bwd").split()
    print(sys.argv[2], sys.argv[3], sys.argv[1], "ms/step", round(d["ms_per_step"], 3), *[f"{k} {round(r[k], 3)} ({d['roofline']['mfma_frac'].get(k)})" for k in keys])
except Exception as e:
    print(sys.argv[1:], "ERR", e, open("gpurun_out/ablib.log").read()[-300:])
PY
    done
  done
done
