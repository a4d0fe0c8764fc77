#!/bin/bash
# round 4, session z: BatchNorm backward in two halves (dy formed in the weight-gradient kernel's staging)
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_lazy_bn.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_graph.py tests/test_gpu_trainer.py tests/test_gpu_fused_ops.py -x -q -m gpu -k "not float64 and not rccl" > gpurun_out/r4z_tests.txt 2>&1
tail -5 gpurun_out/r4z_tests.txt
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 4 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['single_stream']['ms_per_step'], d['config']['final_loss'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['calls_per_step'])[:60]:
    if 'hrwrw' in k: print('   ', k, v)
"; }
for rep in 1 2 3; do
  unset UAPS_LAZY_BN_BWD; run "two halves"
  export UAPS_LAZY_BN_BWD=0; run "one piece"
done 2>&1 | tee gpurun_out/r4z_bench.txt
