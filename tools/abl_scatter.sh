#!/bin/bash
# per-variant atomic-request counts and kernel times of the density scatter (ablation builds abl_*.so)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-render --no-final-stage --exploit-liveness"
for v in base NOXY NOZ L0 L01; do
  if [ $v = base ]; then lib=$PWD/robust-dynrf_amd/librodynrf.so; else lib=$PWD/robust-dynrf_amd/abl_$v.so; fi
  rm -rf /tmp/raw_$v
  RDRF_LIB=$lib rocprofv3 --pmc TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d /tmp/raw_$v -o p -- python $ARGS > gpurun_out/abl_$v.log 2>&1
  RDRF_LIB=$lib python - "$v" <<'PY'
import csv, glob, sys, collections
v = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for fn in glob.glob(f"/tmp/raw_{v}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_scatter" in k:
            a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"] or 0); a[1] += 1
dur = collections.defaultdict(lambda: [0.0, 0])
for fn in glob.glob(f"/tmp/raw_{v}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_scatter" in k:
            d = dur[k]; d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; d[1] += 1
for k in sorted(acc):
    print(v, k, {c: round(s / n) for c, (s, n) in acc[k].items()}, "avg_us", round(dur[k][0] / max(dur[k][1], 1), 1), "n", dur[k][1])
PY
done
