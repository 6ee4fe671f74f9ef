#!/bin/bash
# timing experiment: k_dw3's DMA as full 128-byte rows (same bytes, WRONG arithmetic) against its 64-byte half rows: DMA only
# (NOMFMA variants) and the whole kernel -> is the half-row access pattern what holds the memory side at 4.8 TB/s?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
{
for stage in stage0 final; do for i in 1 2; do for lib in abl_dwnm abl_dwfr librodynrf_tools abl_dwfrfull; do
  RDRF_LIB=$PWD/robust-dynrf_amd/$lib.so timeout 300 python bench.py --stage $stage --steps 6 --warmup 2 --no-cpu-baseline --no-final-stage --no-render --no-sparse --no-liveness-leg >/dev/null 2>&1
  python - "$stage $lib" <<'PY'
import json, sys
d = json.load(open("bench_detail.json")); r = d["roofline"]["kernel_ms_per_step"]
print(sys.argv[1], {k: round(v, 3) for k, v in r.items() if k.startswith("dw_")})
PY
done; done; done
} > gpurun_out/dw_fullrow.txt 2>&1
cat gpurun_out/dw_fullrow.txt
