#!/bin/bash
# round 4, session x: full-width-row kernels on 256-wide column strips (W = 512, 768)
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fused_ops.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu -k "not float64 and not rccl" > gpurun_out/r4x_tests.txt 2>&1
tail -5 gpurun_out/r4x_tests.txt
python bench.py --net unet_uaps --in-chns 1 --classes 2 --aux 5 --batch 8 --size 512 --steps 8 --warmup 4 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 2 2>gpurun_out/r4x_bench.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs3', d['value'], d['ms_per_step'], d['config'].get('final_loss'))
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['calls_per_step'])[:16]: print('   %8.1f us/step' % (v['avg_us']*v['calls_per_step']), k, v)
" | tee gpurun_out/r4x_bench.txt
UAPS_DIAG_NO_ROW16=1 UAPS_DIAG_NO_ROW_WRW=1 python bench.py --net unet_uaps --in-chns 1 --classes 2 --aux 5 --batch 8 --size 512 --steps 8 --warmup 4 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs3 tile kernels', d['value'], d['ms_per_step'], d['config'].get('final_loss'))
" | tee -a gpurun_out/r4x_bench.txt
