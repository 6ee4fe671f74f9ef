#!/bin/bash
# interleaved A/B of environment-switch combinations on the PRODUCT library (python-level switches):
#   tools/ab_pyenv.sh "<bench args>" "A=0" "A=1" ...   prints the per-kernel table (> 0.1 ms)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ARGS=$1; shift
for i in 1 2; do for combo in "$@"; do
  env $combo timeout 300 python bench.py --full-line $ARGS --no-cpu-baseline --no-final-stage --no-render --no-sparse 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['kernel_ms_per_step']
print('$combo', '| ms/step', round(d['ms_per_step'],3), 'rays/s', round(d['value']), {k: round(v,3) for k,v in r.items() if v > 0.1})"
done; done
