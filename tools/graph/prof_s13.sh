#!/bin/bash
# kernel-time table of the launch-bound stages (S = 13): where the GPU time of a captured iteration goes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
ulimit -c 0
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
ROOT=$PWD
for c in nvidia_no_poses davis; do
  RAW=/tmp/raw_s13_$c; rm -rf $RAW; mkdir -p $RAW
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o t -- python $ROOT/bench.py --config $c --stage stage0 --steps 40 --warmup 6 --no-cpu-baseline --no-roofline --no-sparse --no-final-stage --no-render --no-liveness-leg ${S13_EXTRA} > $OUT/s13_$c.log 2>&1 )
  cp $(find $RAW -name "*kernel_stats.csv" | head -1) gpurun_out/s13_${c}_kernel_stats.csv
  python - gpurun_out/s13_${c}_kernel_stats.csv 46 <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(sys.argv[1], "total kernel ms/step", round(tot / steps / 1e6, 3), "launches/step", round(sum(int(r["Calls"]) for r in rows) / steps, 1))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print(f'{float(r["TotalDurationNs"]) / steps / 1e3:8.1f} us/step {int(r["Calls"]) / steps:6.1f} calls/step {float(r["AverageNs"]) / 1e3:7.1f} us  {r["Name"][:90]}')
PY
done
