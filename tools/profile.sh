#!/bin/bash
# rocprofv3 profile of the bench on the GPU box (run through gpurun). Writes small per-kernel
# summaries under gpurun_out/prof/ (raw traces are deleted: gpurun_out is capped at 64 MiB).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
RAW=/tmp/rawprof
rm -rf $OUT $RAW; mkdir -p $OUT $RAW
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-sparse $BENCH_EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/fetch -o fetch -- python $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/write -o write -- python $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $RAW/mfma -o mfma -- python $ARGS > $OUT/mfma.log 2>&1
find $RAW -name "*.csv" -exec ls -la {} \;
cp $(find $RAW/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
python tools/summarize_pmc.py $RAW $OUT
ls -la $OUT
