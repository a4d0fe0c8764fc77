#!/bin/bash
# round 4, session i: 2-deep prefetch in the row weight gradient
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "forward_backward or cat_equals" > gpurun_out/r4i_tests.txt 2>&1
tail -3 gpurun_out/r4i_tests.txt
for b in 32 128; do echo "=== batch $b"; python tools/bench_conv.py --batch $b --no-miopen --only up4a,up4b,dec.out 2>/dev/null; done > gpurun_out/r4i_bench_conv.txt 2>&1
cat gpurun_out/r4i_bench_conv.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 4 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['single_stream']['ms_per_step'], d['config']['final_loss'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['calls_per_step'])[:16]: print('   ', k, v)
" | tee gpurun_out/r4i_bench.txt
