#!/bin/bash
# step-level A/B of process-level knobs on one box: default / HIP_FORCE_DEV_KERNARG=1 / the iteration captured as a HIP graph (stage 0 and final)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out; : > gpurun_out/env_ab.txt
COMMON="--full-line --config nvidia --steps 60 --warmup 10 --no-cpu-baseline --no-final-stage --no-render --no-sparse --no-graph-leg --no-liveness-leg --no-roofline"
for stage in stage0 final; do
  for r in 1 2 3; do
    for v in default kernarg graph; do
      case $v in
        default) E=""; X="";;
        kernarg) E="HIP_FORCE_DEV_KERNARG=1"; X="";;
        graph) E=""; X="--graph";;
      esac
      ms=$(env $E timeout 300 python bench.py $COMMON --stage $stage $X 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))" 2>&1 | tail -1)
      echo "$stage $r $v ms/step $ms" | tee -a gpurun_out/env_ab.txt
    done
  done
done
