#!/bin/bash
# PMC passes over tools/fwd_ab.py (forward field kernels on a fixed workload): instruction-cache, wait and MFMA counters per
# kernel -> gpurun_out/pmc_fwd.csv.  Separate --pmc passes, kernel-trace only (the pool's rule).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
RAW=/tmp/rawfwd
rm -rf $RAW; mkdir -p $RAW gpurun_out
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  ( cd /tmp && rocprofv3 --pmc $line --kernel-trace --output-format csv -d $RAW/p$i -o p -- python $GRAFT_REPO_ROOT/tools/fwd_ab.py > $RAW/p$i.log 2>&1 )
  tail -1 $RAW/p$i.log
done <<'EOL'
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_VALU
SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU
EOL
python - "$RAW" <<'PY'
import csv, glob, os, sys, collections
raw = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for fn in glob.glob(os.path.join(raw, "**", "*counter_collection.csv"), recursive=True):
    with open(fn) as f:
        for r in csv.DictReader(f):
            k = r.get("Kernel_Name", "?").split("(")[0].replace(",", ";")
            a = acc[k][r.get("Counter_Name")]; a[0] += float(r.get("Counter_Value", 0) or 0); a[1] += 1
with open("gpurun_out/pmc_fwd.csv", "w") as f:
    f.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for k in sorted(acc):
        if "k_" not in k: continue
        for c, (s, n) in sorted(acc[k].items()):
            f.write(f"{k},{c},{n},{s/n:.6g}\n")
print("kernels", len(acc))
PY
