#!/bin/bash
# round 4: BatchNorm backward in two halves on the 256-wide decoder layers against the one-piece backward, same box, interleaved
export TMPDIR=/tmp
run() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['step_ms']['p50'], d['config']['final_loss'])
"; }
for rep in 1 2 3 4 5 6; do
  unset UAPS_LAZY_BN_BWD; run "two halves"
  export UAPS_LAZY_BN_BWD=0; run "one piece"
done 2>&1 | tee gpurun_out/r4z3_ab.txt
