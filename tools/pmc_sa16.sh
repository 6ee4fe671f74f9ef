#!/bin/bash
# PMC A/B of the static appearance kernel: 32-sample tiles (RDRF_SA16=0, k_static_app) vs 16-sample tiles (RDRF_SA16=1,
# k_static_app16) on tools/fwd_ab.py's fixed workload -> gpurun_out/pmc_sa16.csv   (usage: tools/pmc_sa16.sh [N S grid])
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
export RDRF_LIB=$PWD/robust-dynrf_amd/librodynrf_tools.so
ARGS="${1:-16384} ${2:-270} ${3:-331,368,220}"
RAW=/tmp/rawsa16
rm -rf $RAW; mkdir -p $RAW gpurun_out
for x in 0 1; do
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  ( cd /tmp && RDRF_SA16=$x rocprofv3 --pmc $line --kernel-trace --output-format csv -d $RAW/v$x/p$i -o p -- python $GRAFT_REPO_ROOT/tools/fwd_ab.py $ARGS > $RAW/v$x.p$i.log 2>&1 )
  tail -1 $RAW/v$x.p$i.log | cut -c1-160
done <<'EOL'
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_VALU
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE
EOL
done
python - "$RAW" <<'PY'
import csv, glob, os, sys, collections
raw = sys.argv[1]
with open("gpurun_out/pmc_sa16.csv", "w") as f:
    f.write("variant,kernel,counter,dispatches,mean_per_dispatch\n")
    for x in ("0", "1"):
        acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for fn in glob.glob(os.path.join(raw, "v" + x, "**", "*counter_collection.csv"), recursive=True):
            with open(fn) as g:
                for r in csv.DictReader(g):
                    k = r.get("Kernel_Name", "?").split("(")[0].replace(",", ";")
                    a = acc[k][r.get("Counter_Name")]; a[0] += float(r.get("Counter_Value", 0) or 0); a[1] += 1
        for k in sorted(acc):
            if "k_static_app" not in k: continue
            for c, (s, n) in sorted(acc[k].items()):
                f.write(f"RDRF_SA16={x},{k},{c},{n},{s/n:.6g}\n")
PY
cat gpurun_out/pmc_sa16.csv
