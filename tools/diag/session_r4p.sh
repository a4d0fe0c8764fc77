#!/bin/bash
# round 4, session p: first-layer weight gradient on the row kernel (input bound from the batch concatenation); advisor items
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fused_ops.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_graph.py tests/test_res_uaps.py tests/test_gpu_strided.py -q -m gpu -k "not float64 and not rccl" > gpurun_out/r4p_tests.txt 2>&1
tail -6 gpurun_out/r4p_tests.txt
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 4 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['single_stream']['ms_per_step'], d['config']['final_loss'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['calls_per_step'])[:60]:
    if 'wrw' in k and ('hrwrw' in k or 'conv_wrw_kernel' in k): print('   ', k, v)
"; }
for rep in 1; do
  unset UAPS_DIAG_NO_ROW_WRW; run "row wrw"
done 2>&1 | tee gpurun_out/r4p_bench.txt
