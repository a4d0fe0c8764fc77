#!/bin/bash
# round 4, session b: hp16 staging fix (correctness + speed), stamp table
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not float64" > gpurun_out/r4b_tests.txt 2>&1
tail -3 gpurun_out/r4b_tests.txt
for b in 32 128; do echo "=== batch $b"; python tools/bench_conv.py --batch $b --no-miopen --only up4,dec.out,enc.in 2>/dev/null; done > gpurun_out/r4b_bench_conv.txt 2>&1
cat gpurun_out/r4b_bench_conv.txt
timeout 900 python tools/diag/stamp_table.py > gpurun_out/r4b_stamp_table.txt 2>&1
cat gpurun_out/r4b_stamp_table.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --other-configs 0 --exact-steps 0 > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r4b_bench.json')); print(d['value'], d['ms_per_step'], d['single_stream'], d['roofline']['kernel'], d['roofline']['avg_us'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['calls_per_step'])[:25]: print(k, v)
"
