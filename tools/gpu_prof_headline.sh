#!/bin/bash
# rocprofv3 kernel-trace stats of the HEADLINE mode of bench.py (captured hipGraph replayed per step, one HIP stream per auxiliary
# decoder).  Kernels of different decoders overlap here, so a kernel's duration includes its share of a contended chip and
# the durations sum to more than the step: the per-kernel roofline figures come from tools/gpu_prof.sh (single stream).
set -x
mkdir -p gpurun_out/prof_headline
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_headline -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --analysis-steps 0 --exact-steps 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_headline/bench_stdout.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_headline -name "*kernel_stats.csv" | head -1)
head -12 "$f"
tail -2 gpurun_out/prof_headline/bench_stdout.log | cut -c1-300
find gpurun_out/prof_headline -name "*kernel_trace.csv" -size +20M -delete
