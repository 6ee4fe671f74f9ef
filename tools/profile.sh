#!/bin/bash
# rocprofv3 profile of the bench on the GPU box (run through gpurun). Writes gpurun_out/prof_*.
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o trace -- python $ARGS > $OUT/prof_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_fetch -o fetch -- python $ARGS > $OUT/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_write -o write -- python $ARGS > $OUT/prof_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/prof_mfma -o mfma -- python $ARGS > $OUT/prof_mfma.log 2>&1
find $OUT -name "*.csv" | head -30
ls -la $OUT/prof_trace/* | head
