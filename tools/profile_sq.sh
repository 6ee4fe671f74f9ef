#!/bin/bash
# SQ instruction-mix / stall counters per kernel (separate --pmc passes; kernel-trace only).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/sq
RAW=/tmp/rawsq
rm -rf $OUT $RAW; mkdir -p $OUT $RAW
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-render --no-sparse $BENCH_EXTRA"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line --kernel-trace --output-format csv -d $RAW/p$i -o p -- python $ARGS > $OUT/p$i.log 2>&1
  tail -2 $OUT/p$i.log
done <<'EOL'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_BRANCH
TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
EOL
python - "$RAW" "$OUT" <<'PY'
import csv, glob, os, sys, collections
raw, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for fn in glob.glob(os.path.join(raw, "**", "*counter_collection.csv"), recursive=True):
    with open(fn) as f:
        for r in csv.DictReader(f):
            k = r.get("Kernel_Name", "?").split("(")[0].replace(",", ";")
            a = acc[k][r.get("Counter_Name")]; a[0] += float(r.get("Counter_Value", 0) or 0); a[1] += 1
with open(os.path.join(out, "sq.csv"), "w") as f:
    f.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for k in sorted(acc):
        if not ("k_" in k): continue
        for c, (s, n) in sorted(acc[k].items()):
            f.write(f"{k},{c},{n},{s/n:.6g}\n")
print("kernels", len(acc))
PY
ls -la $OUT
