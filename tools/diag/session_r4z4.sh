#!/bin/bash
# round 4: the two-halves BatchNorm backward on the column-strip kernels: tests, then configs[3] with and without it, same box
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_lazy_bn.py tests/test_gpu_parity.py -x -q -m gpu -k "not rccl" > gpurun_out/r4z4_tests.txt 2>&1
tail -3 gpurun_out/r4z4_tests.txt
run() { python bench.py --net unet_uaps --in-chns 1 --classes 2 --aux 5 --batch 8 --size 512 --steps 10 --warmup 4 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs3 $1', d['value'], d['ms_per_step'], d['config'].get('final_loss'))
"; }
for rep in 1 2 3; do
  unset UAPS_LAZY_BN_BWD; run "two halves"
  export UAPS_LAZY_BN_BWD=0; run "one piece"
done 2>&1 | tee gpurun_out/r4z4_ab.txt
