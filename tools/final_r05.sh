#!/bin/bash
# Round-5 closing artifacts from ONE gpurun call on one box: the PMC passes of the headline configuration and of the two other
# configurations (their summaries feed the bench line's `traffic` / `step_hbm_bytes`), the default bench line (headline, analysis pass
# with the library-side algorithmic-byte tally, strict-arithmetic legs, inference block, other configs, CPU baseline), single-stream
# kernel stats of the headline step and of the ResNet-50 configuration, and the N > 1 code path on one card (four gloo ranks, 2 host
# cores per rank, eager against two-graph).  Copy gpurun_out/r05/* into profiles/ afterwards.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
cd $R
bash tools/gpu_pmc.sh > gpurun_out/r05/pmc.log 2>&1
cp gpurun_out/pmc/summary.txt gpurun_out/r05/pmc_traffic_summary.txt
cp gpurun_out/pmc/pmc_traffic.json gpurun_out/r05/pmc_traffic.json
bash tools/gpu_pmc_cfg.sh configs3 --net unet_uaps --in-chns 1 --classes 2 --aux 5 --batch 8 --size 512 > gpurun_out/r05/pmc_configs3.log 2>&1
bash tools/gpu_pmc_cfg.sh configs4 --net resnet50_uaps --in-chns 3 --classes 2 --aux 3 --batch 8 --size 640 > gpurun_out/r05/pmc_configs4.log 2>&1
for t in configs3 configs4; do
  cp gpurun_out/pmc_$t/summary.txt gpurun_out/r05/pmc_traffic_summary_$t.txt
  cp gpurun_out/pmc_$t/pmc_traffic.json gpurun_out/r05/pmc_traffic_$t.json
  cp gpurun_out/pmc_$t/pmc_traffic.json profiles/pmc_traffic_$t.json      # the box's copy: the bench line below reads these
done
cp gpurun_out/pmc/pmc_traffic.json profiles/pmc_traffic.json
timeout 1500 python bench.py > gpurun_out/r05/bench.json 2> gpurun_out/r05/bench.err
tail -c 400 gpurun_out/r05/bench.err
bash tools/gpu_prof.sh > gpurun_out/r05/prof.log 2>&1
cp $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) gpurun_out/r05/kernel_stats.csv
bash tools/diag/prof_resnet.sh > gpurun_out/r05/prof_resnet.log 2>&1
cp $(find gpurun_out/prof_res -name "*kernel_stats.csv" | head -1) gpurun_out/r05/resnet50_kernel_stats.csv
bash tools/diag/n4_gloo_bench.sh > gpurun_out/r05/n4_one_card_gloo.txt 2>&1
cat gpurun_out/r05/n4_one_card_gloo.txt
ls -la gpurun_out/r05
