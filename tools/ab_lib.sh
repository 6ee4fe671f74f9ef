#!/bin/bash
# per-kernel table of the bench for several builds: tools/ab_lib.sh lib1.so lib2.so ...  (paths relative to the repo)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
for lib in "$@"; do
  RDRF_LIB=$PWD/$lib timeout 300 python bench.py --full-line --steps 30 --warmup 5 --no-cpu-baseline --no-final-stage --no-render --no-sparse $BENCH_EXTRA 2>&1 | tail -1 > gpurun_out/ablib.log
  python - "$lib" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/ablib.log").read().strip().splitlines()[-1]); r = d["roofline"]["kernel_ms_per_step"]
    print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), "sum", round(d["roofline"]["sum_kernel_ms_per_step"], 2), {k: round(v, 3) for k, v in r.items() if v > 0.1})
except Exception as e:
    print(sys.argv[1], "ERR", e, open("gpurun_out/ablib.log").read()[-500:])
PY
done
