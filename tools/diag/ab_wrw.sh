#!/bin/bash
# A/B of the weight-gradient side streams (uaps_amd/conv.py: WRW_STREAMS) on one box: headline mode, interleaved.
export TMPDIR=/tmp
OUT=gpurun_out/ab_wrw.txt
: > $OUT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "side_streams or decoder_streams_give" 2>&1 | tail -5 | tee -a $OUT
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --other-configs 0 --analysis-steps 0 --exact-steps 0"
one() { echo "== $1" | tee -a $OUT; shift; env "$@" 2>>gpurun_out/ab_wrw.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms'], d['config']['launch_mode'][:60])" | tee -a $OUT; }
for rep in 1 2; do
  one "wrw=0 graph" UAPS_WRW_STREAMS=0 $B
  one "wrw=1 graph" UAPS_WRW_STREAMS=1 $B
  one "wrw=1 graph hwq8" UAPS_WRW_STREAMS=1 GPU_MAX_HW_QUEUES=8 $B
  one "wrw=0 graph hwq8" UAPS_WRW_STREAMS=0 GPU_MAX_HW_QUEUES=8 $B
done
one "wrw=0 eager" UAPS_WRW_STREAMS=0 $B --no-graph
one "wrw=1 eager" UAPS_WRW_STREAMS=1 $B --no-graph
one "wrw=1 eager hwq8" UAPS_WRW_STREAMS=1 GPU_MAX_HW_QUEUES=8 $B --no-graph
one "single-stream graph" UAPS_WRW_STREAMS=0 $B --single-stream
