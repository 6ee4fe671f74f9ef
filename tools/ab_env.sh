#!/bin/bash
# interleaved A/B of one environment switch on the training leg: tools/ab_env.sh VAR [steps]   (VAR=0 / VAR=1, twice each)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
V=$1; K=${2:-40}
for i in 1 2; do for x in 0 1; do
  env $V=$x timeout 300 python bench.py --full-line --steps $K --warmup 5 --no-cpu-baseline --no-final-stage --no-render --no-sparse 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['kernel_ms_per_step']
print('$V=$x', 'ms/step', round(d['ms_per_step'],3), 'rays/s', round(d['value']), {k: round(v,3) for k,v in r.items() if k in ('dyn_density','dyn_app','static_app','composite_bwd')})"
done; done
