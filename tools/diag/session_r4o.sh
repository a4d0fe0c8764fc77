#!/bin/bash
# round 4, session o: BatchNorm backward group by group (dx finds in the memory-side cache what sums just read); A/B on one box
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_fused_ops.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not float64 and not rccl" > gpurun_out/r4o_tests.txt 2>&1
tail -4 gpurun_out/r4o_tests.txt
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 4 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['single_stream']['ms_per_step'], d['config']['final_loss'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['calls_per_step'])[:40]:
    if 'bn_bwd' in k: print('   ', k, v)
"; }
for rep in 1 2 3; do
  unset UAPS_DIAG_NO_BN_GROUP_SPLIT; run "by group"
  export UAPS_DIAG_NO_BN_GROUP_SPLIT=1; run "whole batch"
done 2>&1 | tee gpurun_out/r4o_bench.txt
