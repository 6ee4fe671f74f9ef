#!/bin/bash
# final-stage step (grid [331,368,220], S=270): kernel stats + fetch / write / mfma PMC passes -> gpurun_out/prof5f/
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof5f
RAW=/tmp/rawprof5f
rm -rf $OUT $RAW; mkdir -p $OUT $RAW
TRAIN="$PWD/bench.py --stage final --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-sparse --no-final-stage --no-render"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/train -o t -- python $TRAIN > $OUT/train.log 2>&1 )
cp $(find $RAW/train -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
while read -r tag line; do
  [ -z "$line" ] && continue
  ( cd /tmp && rocprofv3 --pmc $line --kernel-trace --output-format csv -d $RAW/$tag -o p -- python $TRAIN > $OUT/$tag.log 2>&1 )
done <<'EOL'
fetch FETCH_SIZE
write WRITE_SIZE
mfma SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
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
agg(["fetch"], "pmc_fetch.csv"); agg(["write"], "pmc_write.csv"); agg(["mfma"], "pmc_mfma.csv")
PY
ls -la $OUT
