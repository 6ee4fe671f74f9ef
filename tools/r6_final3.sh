#!/bin/bash
# round 6, final library: the complete suite with the margin of every tolerance check, the seed sweep, the 1500-iteration soak run
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out; rm -f /tmp/m.tsv
RDRF_MARGINS=/tmp/m.tsv python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/suite_run14.txt; cat gpurun_out/suite_run14.txt
python tools/margins.py /tmp/m.tsv 25 > gpurun_out/parity_margins.txt; head -6 gpurun_out/parity_margins.txt
timeout 600 python tools/seed_sweep.py > gpurun_out/seed_sweep.txt 2>&1; tail -3 gpurun_out/seed_sweep.txt
timeout 600 python tools/long_run.py 1500 > gpurun_out/long_run.txt 2>&1; tail -4 gpurun_out/long_run.txt
