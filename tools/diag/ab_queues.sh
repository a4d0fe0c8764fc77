#!/bin/bash
# How much launch concurrency pays: hardware queues (GPU_MAX_HW_QUEUES) under the headline stream mode (one stream per auxiliary
# decoder), one box.  Round 4 also tried a companion stream per decoder stream for the weight-gradient launches (8 streams):
# 14.13 -> 14.8-15.0 ms on 4 hardware queues, 20.7 ms on 8 -- more concurrent kernels thrash each other; the mechanism was removed.
export TMPDIR=/tmp
OUT=gpurun_out/ab_queues.txt
: > $OUT
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --other-configs 0 --analysis-steps 0 --exact-steps 0"
one() { echo "== $1" | tee -a $OUT; shift; env "$@" 2>>gpurun_out/ab_queues.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms']['p50'])" | tee -a $OUT; }
for q in 1 2 3 4 8; do
  one "decoder streams, hwq=$q" GPU_MAX_HW_QUEUES=$q $B
done
one "single stream" $B --single-stream
