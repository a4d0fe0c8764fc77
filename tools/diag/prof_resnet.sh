#!/bin/bash
# rocprofv3 kernel stats of the ResNet-50 configuration (GPU box)
mkdir -p gpurun_out/prof_res
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_res -o bench -- python3 $R/bench.py --net resnet50_uaps --size 640 --classes 2 --batch 8 --steps 4 --warmup 2 --analysis-steps 0 --exact-steps 0 --single-stream --no-graph --no-cpu-baseline --no-inference > $R/gpurun_out/prof_res/stdout.log 2>&1
cd $R
f=$(find gpurun_out/prof_res -name "*kernel_stats.csv" | head -1)
head -45 "$f" | cut -c1-200
find gpurun_out/prof_res -name "*kernel_trace.csv" -size +20M -delete
