#!/bin/bash
# rocprofv3 kernel-trace stats of the bench command (GPU box).
set -x
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --analysis-steps 0 --exact-steps 0 --single-stream --no-graph --no-cpu-baseline --no-inference --other-configs 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/bench_stdout.log 2>&1
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/prof | head
find gpurun_out/prof -name "*stats*" | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
head -40 "$f"
# keep only the small stats files for merge-back
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
