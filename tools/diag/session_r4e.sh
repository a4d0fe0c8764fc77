#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "side_streams" > gpurun_out/r4e_tests.txt 2>&1
tail -3 gpurun_out/r4e_tests.txt
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export UAPS_DIAG_NO_ROW16=1; else unset UAPS_DIAG_NO_ROW16; fi
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 4 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NO_ROW16=$v', d['value'], d['ms_per_step'], d['single_stream']['ms_per_step'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['calls_per_step'])[:12]: print('   ', k, v)
"
done 2>&1 | tee gpurun_out/r4e_bench.txt
