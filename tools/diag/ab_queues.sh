#!/bin/bash
# How much concurrency pays: hardware queues x stream modes, headline protocol, one box.
export TMPDIR=/tmp
OUT=gpurun_out/ab_queues.txt
: > $OUT
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --other-configs 0 --analysis-steps 0 --exact-steps 0"
one() { echo "== $1" | tee -a $OUT; shift; env "$@" 2>>gpurun_out/ab_queues.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms']['p50'])" | tee -a $OUT; }
for q in 1 2 3 4; do
  one "decoder streams, wrw=0, hwq=$q" UAPS_WRW_STREAMS=0 GPU_MAX_HW_QUEUES=$q $B
done
one "decoder streams off, wrw=1, hwq=4" UAPS_WRW_STREAMS=1 UAPS_BENCH_DECODER_STREAMS=0 $B
one "decoder streams off, wrw=1, hwq=2" UAPS_WRW_STREAMS=1 UAPS_BENCH_DECODER_STREAMS=0 GPU_MAX_HW_QUEUES=2 $B
one "decoder streams, wrw=1, hwq=2" UAPS_WRW_STREAMS=1 GPU_MAX_HW_QUEUES=2 $B
one "decoder streams, wrw=0, default" UAPS_WRW_STREAMS=0 $B
