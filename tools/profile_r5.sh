#!/bin/bash
# Round-5 profiles (run through gpurun): rocprofv3 summaries of
#   (1) TRAINING only, the driver's window (bench.py --steps 20 --warmup 5, no render legs): kernel stats + PMC passes
#   (2) RENDER only (tools/render_bench.py: whole frames, then 512-ray chunks): kernel stats
# -> gpurun_out/prof5/*.csv (copy the ones to keep to profiles/r05_*).  Separate --pmc passes, kernel-trace only.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof5
RAW=/tmp/rawprof5
rm -rf $OUT $RAW; mkdir -p $OUT $RAW
TRAIN="$PWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-sparse --no-final-stage --no-render"
RENDER="$PWD/tools/render_bench.py"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/train -o t -- python $TRAIN > $OUT/train.log 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/render -o t -- python $RENDER whole > $OUT/render_whole.log 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/render512 -o t -- python $RENDER chunk512 > $OUT/render_chunk512.log 2>&1 )
cp $(find $RAW/train -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
cp $(find $RAW/render -name "*kernel_stats.csv" | head -1) $OUT/render_kernel_stats.csv 2>/dev/null
cp $(find $RAW/render512 -name "*kernel_stats.csv" | head -1) $OUT/render_chunk512_kernel_stats.csv 2>/dev/null
i=0
while read -r tag line; do
  [ -z "$line" ] && continue
  ( cd /tmp && rocprofv3 --pmc $line --kernel-trace --output-format csv -d $RAW/$tag -o p -- python $TRAIN > $OUT/$tag.log 2>&1 )
  tail -1 $OUT/$tag.log | cut -c1-200
done <<'EOL'
fetch FETCH_SIZE
write WRITE_SIZE
mfma SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS
sq3 TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
EOL
python - "$RAW" "$OUT" <<'PY'
import csv, glob, os, sys, collections
raw, out = sys.argv[1], sys.argv[2]
def agg(tags, name):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for tag in tags:
        for fn in glob.glob(os.path.join(raw, tag, "**", "*counter_collection.csv"), recursive=True):
            with open(fn) as f:
                for r in csv.DictReader(f):
                    k = r.get("Kernel_Name", "?").split("(")[0].replace(",", ";")
                    a = acc[k][r.get("Counter_Name")]; a[0] += float(r.get("Counter_Value", 0) or 0); a[1] += 1
    with open(os.path.join(out, name), "w") as f:
        f.write("kernel,counter,dispatches,mean_per_dispatch,total\n")
        for k in sorted(acc):
            if "k_" not in k: continue
            for c, (s, n) in sorted(acc[k].items()):
                f.write(f"{k},{c},{n},{s/n:.6g},{s:.6g}\n")
    print(name, len(acc), "kernels")
agg(["fetch"], "pmc_fetch.csv"); agg(["write"], "pmc_write.csv"); agg(["mfma"], "pmc_mfma.csv"); agg(["sq1", "sq2", "sq3"], "sq_counters.csv")
PY
ls -la $OUT
