#!/bin/bash
# Round-6 profile of ONE bench.py training configuration (run through gpurun):
#   tools/profile_r6.sh <tag> [sq] -- <bench.py arguments>
# -> gpurun_out/prof6_<tag>/{kernel_stats.csv, pmc_fetch.csv, pmc_write.csv, pmc_mfma.csv[, sq_counters.csv], hbm_table.txt}
# rocprofv3 kernel stats in one run, every --pmc set in a run of its own (kernel-trace only), as the guide prescribes.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
TAG=$1; shift
SQ=0; if [ "$1" = "sq" ]; then SQ=1; shift; fi
[ "$1" = "--" ] && shift
OUT=$PWD/gpurun_out/prof6_$TAG
RAW=/tmp/rawprof6_$TAG
rm -rf $OUT $RAW; mkdir -p $OUT $RAW
TRAIN="$PWD/bench.py $* --no-cpu-baseline --no-roofline --no-sparse --no-final-stage --no-render --no-liveness-leg --no-graph-leg"
echo "$TRAIN" > $OUT/command.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/train -o t -- python $TRAIN > $OUT/train.log 2>&1 )
cp $(find $RAW/train -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
tail -1 $OUT/train.log | cut -c1-300
{
cat <<'EOL'
fetch FETCH_SIZE
write WRITE_SIZE
mfma SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
EOL
if [ $SQ = 1 ]; then cat <<'EOL'
sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS
sq3 TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
EOL
fi
} | while read -r tag line; do
  [ -z "$line" ] && continue
  ( cd /tmp && rocprofv3 --pmc $line --kernel-trace --output-format csv -d $RAW/$tag -o p -- python $TRAIN > $OUT/$tag.log 2>&1 )
done
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
    if not acc: return
    with open(os.path.join(out, name), "w") as f:
        f.write("kernel,counter,dispatches,mean_per_dispatch,total\n")
        for k in sorted(acc):
            if "k_" not in k: continue
            for c, (s, n) in sorted(acc[k].items()):
                f.write(f"{k},{c},{n},{s/n:.6g},{s:.6g}\n")
    print(name, len(acc), "kernels")
agg(["fetch"], "pmc_fetch.csv"); agg(["write"], "pmc_write.csv"); agg(["mfma"], "pmc_mfma.csv"); agg(["sq1", "sq2", "sq3"], "sq_counters.csv")
PY
python $PWD/tools/hbm_table.py $OUT > $OUT/hbm_table.txt 2>&1
cat $OUT/hbm_table.txt | head -40
