#!/bin/bash
# Round-6 closing artifacts from ONE gpurun call on one box (copy gpurun_out/r06/* into profiles/ afterwards, prefixed r06_final_):
# the same-arithmetic GEMM ceiling (bench.py reads its committed JSON for roofline.practical_peak), the PMC passes of the three
# configurations, the default bench line (headline with clock / power sampling, analysis pass, strict-arithmetic legs, inference block,
# other configs, CPU baseline), single-stream kernel stats of the headline step and of the ResNet-50 configuration, and the loss
# block's batch scaling.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r06/gpu_tests_tail.txt; cat gpurun_out/r06/gpu_tests_tail.txt
timeout 600 tools/bin/split_gemm_ceiling gpurun_out/r06/split_gemm_ceiling.json > gpurun_out/r06/split_gemm_ceiling.txt 2>&1
cp gpurun_out/r06/split_gemm_ceiling.json profiles/r06_split_gemm_ceiling.json      # the box's copy: the bench line below reads it
timeout 600 tools/bin/split_wrw_ceiling gpurun_out/r06/split_wrw_ceiling.json > gpurun_out/r06/split_wrw_ceiling.txt 2>&1
cp gpurun_out/r06/split_wrw_ceiling.json profiles/r06_split_wrw_ceiling.json      # (same: read by bench.py when a weight-gradient tile kernel leads)
bash tools/gpu_pmc.sh > gpurun_out/r06/pmc.log 2>&1
cp gpurun_out/pmc/summary.txt gpurun_out/r06/pmc_traffic_summary.txt
cp gpurun_out/pmc/pmc_traffic.json gpurun_out/r06/pmc_traffic.json
bash tools/gpu_pmc_cfg.sh configs3 --net unet_uaps --in-chns 1 --classes 2 --aux 5 --batch 8 --size 512 > gpurun_out/r06/pmc_configs3.log 2>&1
bash tools/gpu_pmc_cfg.sh configs4 --net resnet50_uaps --in-chns 3 --classes 2 --aux 3 --batch 8 --size 640 > gpurun_out/r06/pmc_configs4.log 2>&1
for t in configs3 configs4; do
  cp gpurun_out/pmc_$t/summary.txt gpurun_out/r06/pmc_traffic_summary_$t.txt
  cp gpurun_out/pmc_$t/pmc_traffic.json gpurun_out/r06/pmc_traffic_$t.json
  cp gpurun_out/pmc_$t/pmc_traffic.json profiles/pmc_traffic_$t.json      # the box's copy: the bench line below reads these
done
cp gpurun_out/pmc/pmc_traffic.json profiles/pmc_traffic.json
timeout 1500 python bench.py > gpurun_out/r06/bench.json 2> gpurun_out/r06/bench.err
tail -c 600 gpurun_out/r06/bench.err
bash tools/gpu_prof.sh > gpurun_out/r06/prof.log 2>&1
cp $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) gpurun_out/r06/kernel_stats.csv
bash tools/diag/prof_resnet.sh > gpurun_out/r06/prof_resnet.log 2>&1
cp $(find gpurun_out/prof_res -name "*kernel_stats.csv" | head -1) gpurun_out/r06/resnet50_kernel_stats.csv
python tools/diag/loss_batch_scaling.py > gpurun_out/r06/loss_batch_scaling.txt 2>&1
python tools/diag/small_1x1_ab.py > gpurun_out/r06/small_1x1_ab.txt 2>&1
python tools/diag/ceiling_vs_shipped.py gpurun_out/r06/split_gemm_ceiling.json > gpurun_out/r06/ceiling_vs_shipped.txt 2>&1; cat gpurun_out/r06/ceiling_vs_shipped.txt
ls -la gpurun_out/r06
python -c "
import json; d=json.load(open('gpurun_out/r06/bench.json')); print(d['value'], d['ms_per_step'], d['step_ms'], d['single_stream']['ms_per_step']); print({k:v for k,v in d['roofline'].items() if 'note' not in k and k!='practical_peak'}); print(d['roofline'].get('practical_peak',{}).get('tflops')); print(d.get('roofline_loss')); print([ (o.get('ms_per_step'), o.get('images_per_s')) for o in d.get('other_configs',[])]); print(d['cpu_baseline']['value'], d['power']['regions'])"
