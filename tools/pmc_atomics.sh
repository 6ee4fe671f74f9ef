#!/bin/bash
# atomic-request / VALU counters of the scatter kernels for the current env (RDRF_Z_FAST etc.)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
TAG=${1:-x}
RAW=/tmp/rawat_$TAG
rm -rf $RAW; mkdir -p $RAW gpurun_out
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-render --no-final-stage $BENCH_EXTRA"
rocprofv3 --pmc TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_ATOMIC_sum TCC_REQ_sum SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $RAW -o p -- python $ARGS > gpurun_out/pmc_at_$TAG.log 2>&1
python - "$RAW" "$TAG" <<'PY'
import csv, glob, os, sys, collections
raw, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for fn in glob.glob(os.path.join(raw, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r.get("Kernel_Name", "?").split("(")[0]
        if "scatter" not in k: continue
        a = acc[k][r.get("Counter_Name")]; a[0] += float(r.get("Counter_Value", 0) or 0); a[1] += 1
for k in sorted(acc):
    print(tag, k, {c: round(v[0] / max(v[1], 1) / 1e6, 3) for c, v in sorted(acc[k].items())}, "dispatches", max(v[1] for v in acc[k].values()))
PY
