#!/bin/bash
# round 4, session d: full-width-row kernels (conv_hr16): correctness, layer times, step
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_fused_ops.py -x -q -m gpu -k "not float64 and not rccl" > gpurun_out/r4d_tests.txt 2>&1
tail -3 gpurun_out/r4d_tests.txt
for b in 32 128; do echo "=== batch $b"; python tools/bench_conv.py --batch $b --no-miopen --only up4a,up4b,enc.in1 2>/dev/null; done > gpurun_out/r4d_bench_conv.txt 2>&1
echo "=== batch 128, UAPS_DIAG_NO_ROW16=1" >> gpurun_out/r4d_bench_conv.txt
UAPS_DIAG_NO_ROW16=1 python tools/bench_conv.py --batch 128 --no-miopen --only up4a,up4b,enc.in1 2>/dev/null >> gpurun_out/r4d_bench_conv.txt
cat gpurun_out/r4d_bench_conv.txt
for v in 0 1 0 1; do
  UAPS_DIAG_NO_ROW16=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 4 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NO_ROW16=$v', d['value'], d['ms_per_step'], d['single_stream']['ms_per_step'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['calls_per_step'])[:12]: print('   ', k, v)
"
done 2>&1 | tee gpurun_out/r4d_bench.txt
