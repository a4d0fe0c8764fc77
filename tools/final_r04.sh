#!/bin/bash
# Round-4 closing artifacts from ONE gpurun call on one box: the two PMC passes (their summary feeds the bench line's `traffic` and
# `step_hbm_bytes`), the default bench line, single-stream kernel stats, and the kernel stats of the ResNet-50 configuration.
# Copy gpurun_out/r04/* into profiles/ afterwards.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04
cd $R
bash tools/gpu_pmc.sh > gpurun_out/r04/pmc.log 2>&1
cp gpurun_out/pmc/summary.txt gpurun_out/r04/pmc_traffic_summary.txt
cp gpurun_out/pmc/pmc_traffic.json gpurun_out/r04/pmc_traffic.json
cp gpurun_out/pmc/pmc_traffic.json profiles/pmc_traffic.json        # the box's copy: the bench line below reads it
timeout 1500 python bench.py > gpurun_out/r04/bench.json 2> gpurun_out/r04/bench.err
tail -c 400 gpurun_out/r04/bench.err
bash tools/gpu_prof.sh > gpurun_out/r04/prof.log 2>&1
cp $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) gpurun_out/r04/kernel_stats.csv
bash tools/diag/prof_resnet.sh > gpurun_out/r04/prof_resnet.log 2>&1
cp $(find gpurun_out/prof_res -name "*kernel_stats.csv" | head -1) gpurun_out/r04/resnet50_kernel_stats.csv
ls -la gpurun_out/r04
