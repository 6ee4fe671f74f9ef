#!/bin/bash
# round 6: every rocprofv3 summary the DESIGN / bench numbers quote, in one gpurun call -> gpurun_out/prof6_*/ (copy to profiles/r06_*)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/profile_r6.sh stage0 sq -- --steps 20 --warmup 5 > gpurun_out/prof6_stage0.log 2>&1
bash tools/profile_r6.sh final -- --stage final --steps 10 --warmup 3 > gpurun_out/prof6_final.log 2>&1
bash tools/profile_r6.sh 640 -- --config nvidia_no_poses --stage final --steps 6 --warmup 2 > gpurun_out/prof6_640.log 2>&1
bash tools/profile_r6.sh davis_final -- --config davis --stage final --steps 6 --warmup 2 > gpurun_out/prof6_davis_final.log 2>&1
RAW=/tmp/rawprof6_render; rm -rf $RAW; mkdir -p $RAW gpurun_out/prof6_render
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/whole -o t -- python $GRAFT_REPO_ROOT/tools/render_bench.py whole > $GRAFT_REPO_ROOT/gpurun_out/prof6_render/render_whole.txt 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/c512 -o t -- python $GRAFT_REPO_ROOT/tools/render_bench.py chunk512 > $GRAFT_REPO_ROOT/gpurun_out/prof6_render/render_chunk512.txt 2>&1 )
cp $(find $RAW/whole -name "*kernel_stats.csv" | head -1) gpurun_out/prof6_render/render_kernel_stats.csv 2>/dev/null
cp $(find $RAW/c512 -name "*kernel_stats.csv" | head -1) gpurun_out/prof6_render/render_chunk512_kernel_stats.csv 2>/dev/null
bash tools/configs_table.sh > gpurun_out/configs_table.txt 2>&1
for d in stage0 final 640 davis_final; do echo "== $d"; head -14 gpurun_out/prof6_$d/hbm_table.txt; done
cat gpurun_out/configs_table.txt
grep -h "Mpix" gpurun_out/prof6_render/*.txt | tail -4
