#!/bin/bash
# HBM traffic counters of the bench command (GPU box): two separate rocprofv3 --pmc passes, as
# /opt/skills/guides/MI355X_MICROARCH.md (section HBM / rocprofv3 PMC slots) prescribes: FETCH_SIZE and WRITE_SIZE
# do not fit one pass, and no trace domains are combined with --pmc.
set -x
export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pmc
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$c -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --analysis-steps 0 --exact-steps 0 --single-stream --no-graph --no-cpu-baseline --no-inference --other-configs 0 > $GRAFT_REPO_ROOT/gpurun_out/pmc/$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py gpurun_out/pmc 3 > gpurun_out/pmc/summary.txt 2>&1
tail -30 gpurun_out/pmc/summary.txt
# the raw per-dispatch files are large; keep only the summaries for the merge back
find gpurun_out/pmc -name "*counter_collection.csv" -size +8M -delete
