#!/bin/bash
# round 6, final tree (second call): where does the run-to-run variation of d/d xyz through rgb come from -- the atomic accumulation order
# (product build, also with the fp32 appearance layers) or the bf16 x 3 kernels (would show in the deterministic build)?; stage-0 profile
# without the captured-graph leg (rocprofv3 --pmc crashed on it); the driver's bench command; a second complete suite run
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
{
for c in "ndc_relu 777 115" "contract_relu_te 2100 37"; do
  echo "## deterministic build (librodynrf_det.so: same kernels, fixed-point accumulation, sorted lists)"; RDRF_DETERMINISTIC=1 timeout 300 python tools/graph/det_bwd_app.py $c 2>&1 | tail -2
done
echo "## product build with the appearance layers on the fp32 pipe (abl_appf32.so)"; RDRF_LIB=$PWD/robust-dynrf_amd/abl_appf32.so timeout 300 python tools/graph/det_bwd_app.py ndc_relu 777 115 2>&1 | tail -2
echo "## density phase (tools/graph/det_bwd.py), product build"; timeout 300 python tools/graph/det_bwd.py ndc_relu 777 115 2>&1 | tail -1
echo "## density phase, deterministic build"; RDRF_DETERMINISTIC=1 timeout 300 python tools/graph/det_bwd.py ndc_relu 777 115 2>&1 | tail -1
} > gpurun_out/det_bwd_app2.txt 2>&1
cat gpurun_out/det_bwd_app2.txt
export TMPDIR=/tmp
bash tools/profile_r6.sh stage0 sq -- --steps 20 --warmup 5 > gpurun_out/prof6_stage0.log 2>&1
head -16 gpurun_out/prof6_stage0/hbm_table.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_cmd20.log 2>&1
cp bench_detail.json gpurun_out/bench_detail_driver_cmd20.json
tail -1 gpurun_out/bench_driver_cmd20.log | cut -c1-300
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/suite_run13.txt; cat gpurun_out/suite_run13.txt
