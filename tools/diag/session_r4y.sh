#!/bin/bash
# round 4, session y: headline with the library of the commit before the column-strip forms against the current one, same box, interleaved
export TMPDIR=/tmp
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config']['final_loss'])
"; }
for rep in 1 2 3; do
  unset UAPS_HIP_LIB; run "current"
  export UAPS_HIP_LIB=$GRAFT_REPO_ROOT/tools/bin/libuaps_hip_prev.so; run "before strips"
done 2>&1 | tee gpurun_out/r4y_ab.txt
