#!/bin/bash
# Round-3 closing artifacts from ONE gpurun call on one box: the default bench line, single-stream kernel stats, the two PMC
# passes, and the kernel stats of the ResNet-50 configuration.  Copy gpurun_out/r03/* into profiles/ afterwards.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03
cd $R
timeout 1200 python bench.py > gpurun_out/r03/bench.json 2> gpurun_out/r03/bench.err
tail -c 400 gpurun_out/r03/bench.err
bash tools/gpu_prof.sh > gpurun_out/r03/prof.log 2>&1
cp $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) gpurun_out/r03/kernel_stats.csv
bash tools/gpu_pmc.sh > gpurun_out/r03/pmc.log 2>&1
cp gpurun_out/pmc/summary.txt gpurun_out/r03/pmc_traffic_summary.txt
cp gpurun_out/pmc/pmc_traffic.json gpurun_out/r03/pmc_traffic.json
bash tools/diag/prof_resnet.sh > gpurun_out/r03/prof_resnet.log 2>&1
cp $(find gpurun_out/prof_res -name "*kernel_stats.csv" | head -1) gpurun_out/r03/resnet50_kernel_stats.csv
ls -la gpurun_out/r03
