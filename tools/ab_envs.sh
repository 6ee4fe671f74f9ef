#!/bin/bash
# interleaved A/B of environment-switch COMBINATIONS on the tools build (csrc `make tools`: RDRF_ENV lookups live):
#   tools/ab_envs.sh "<bench args>" "A=0 B=0" "A=1 B=0" ...      (each combination twice, interleaved)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
ARGS=$1; shift
for i in 1 2; do for combo in "$@"; do
  env RDRF_LIB=$PWD/robust-dynrf_amd/librodynrf_tools.so $combo timeout 300 python bench.py --full-line $ARGS --no-cpu-baseline --no-final-stage --no-render --no-sparse 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['kernel_ms_per_step']
fam=sum(v for k,v in r.items() if k.startswith('scatter') or k=='sort')
print('$combo', '| ms/step', round(d['ms_per_step'],3), 'rays/s', round(d['value']), 'scatter+sort', round(fam,3), {k: round(v,3) for k,v in r.items() if k.startswith('scatter') or k=='sort'})"
done; done
