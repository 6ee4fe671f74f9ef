#!/bin/bash
# ablation of the dW kernels (tools-build variants): full / without the global loads (NOLOAD) / without the MFMAs (NOMFMA),
# k_dw2 (RDRF_DW3=0) and k_dw3 (RDRF_DW3=1), per-launch HIP-event times of the training step's dw ranges
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
{
for stage in stage0 final; do for lib in librodynrf_tools abl_dwnl abl_dwnm; do for x in 0 1; do
  RDRF_LIB=$PWD/robust-dynrf_amd/$lib.so RDRF_DW3=$x timeout 300 python bench.py --stage $stage --steps 12 --warmup 4 --no-cpu-baseline --no-final-stage --no-render --no-sparse >/dev/null 2>&1
  python - "$stage $lib RDRF_DW3=$x" <<'PY'
import json, sys
d = json.load(open("bench_detail.json")); r = d["roofline"]["kernel_ms_per_step"]
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in r.items() if k.startswith("dw_")}, "dw sum", round(sum(v for k, v in r.items() if k.startswith("dw_")), 3))
PY
done; done; done
} > gpurun_out/dw3_abl.txt 2>&1
cat gpurun_out/dw3_abl.txt
