#!/bin/bash
# round 6, final tree: reproducibility probes of the appearance backward, the driver's bench command, then every profile (tools/r6_profiles.sh)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
{ for c in "contract_relu_te 2100 37" "ndc_relu 777 115"; do timeout 300 python tools/graph/det_bwd_app.py $c 2>&1 | tail -2; done; } > gpurun_out/det_bwd_app.txt
cat gpurun_out/det_bwd_app.txt
( sleep 20; for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr -s ' ' | head -6; echo --; sleep 8; done ) > gpurun_out/clocks_during_bench.txt 2>&1 &
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_cmd20.log 2>&1
cp bench_detail.json gpurun_out/bench_detail_driver_cmd20.json
tail -1 gpurun_out/bench_driver_cmd20.log | cut -c1-400
wait
bash tools/r6_profiles.sh 2>&1 | tail -60
