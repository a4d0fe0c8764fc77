#!/bin/bash
# round 4, session f: whole GPU suite + bench with hr16 for out_conv + kernel stats
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r4f_tests.txt 2>&1
tail -5 gpurun_out/r4f_tests.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 4 2>/dev/null > gpurun_out/r4f_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r4f_bench.json')); print(d['value'], d['ms_per_step'], d['single_stream']['ms_per_step'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['calls_per_step'])[:30]: print('   ', k, v)
"
bash tools/gpu_prof.sh > gpurun_out/r4f_prof.log 2>&1
cp $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) gpurun_out/r4f_kernel_stats.csv
