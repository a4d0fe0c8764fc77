#!/bin/bash
# round 4, session g: row weight-gradient kernels
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "forward_backward or cat_equals or statistics" > gpurun_out/r4g_tests.txt 2>&1
tail -5 gpurun_out/r4g_tests.txt
for b in 32 128; do echo "=== batch $b"; python tools/bench_conv.py --batch $b --no-miopen --only up4a,up4b,enc.in1,dec.out 2>/dev/null; done > gpurun_out/r4g_bench_conv.txt 2>&1
cat gpurun_out/r4g_bench_conv.txt
for v in 0 1 0; do
  if [ $v = 1 ]; then export UAPS_DIAG_NO_ROW16=1; else unset UAPS_DIAG_NO_ROW16; fi
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 4 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NO_ROW16=$v', d['value'], d['ms_per_step'], d['single_stream']['ms_per_step'], d['config']['final_loss'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['calls_per_step'])[:14]: print('   ', k, v)
"
done 2>&1 | tee gpurun_out/r4g_bench.txt
