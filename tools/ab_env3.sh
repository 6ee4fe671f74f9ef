#!/bin/bash
# A/B/C of env settings on ONE box: tools/ab_env3.sh "<env A>" "<env B>" "<env C>" [bench args]; 2 interleaved rounds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
# the RDRF_* experiment switches of the C library exist in the tools build only: make -C robust-dynrf_amd/csrc tools
[ -f robust-dynrf_amd/librodynrf_tools.so ] && export RDRF_LIB=${RDRF_LIB:-$PWD/robust-dynrf_amd/librodynrf_tools.so}
A="$1"; B="$2"; Cc="$3"; shift 3
mkdir -p gpurun_out
for r in 1 2; do
  for arm in A B C; do
    case $arm in A) E="$A";; B) E="$B";; C) E="$Cc";; esac
    env $E timeout 300 python bench.py --full-line --steps 40 --warmup 5 --no-cpu-baseline --no-final-stage --no-render --no-sparse "$@" 2>&1 | tail -1 > gpurun_out/ab_${arm}_$r.log
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_[ABC]_[12].log")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]["kernel_ms_per_step"]
        print(f, "ms/step", round(d["ms_per_step"], 3), "liveness", round(d["liveness_exploited"]["ms_per_step"], 3),
              "sum_kernels", round(d["roofline"]["sum_kernel_ms_per_step"], 2), {k: round(v, 3) for k, v in r.items() if v > 0.1})
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-800:])
PY
