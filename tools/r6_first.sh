#!/bin/bash
# round 6, first GPU call: the driver's bench command on the unchanged kernels (does the compact line parse?), then the
# profiles of the two configurations whose factors exceed the caches
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_cmd.log 2> gpurun_out/bench_driver_cmd.err
tail -c 3000 gpurun_out/bench_driver_cmd.log
cp bench_detail.json gpurun_out/bench_detail_driver_cmd.json
bash tools/profile_r6.sh 640 -- --config nvidia_no_poses --stage final --steps 6 --warmup 2
bash tools/profile_r6.sh davis_final -- --config davis --stage final --steps 6 --warmup 2
