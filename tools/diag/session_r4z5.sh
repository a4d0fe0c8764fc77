#!/bin/bash
export TMPDIR=/tmp
run() { python bench.py --net unet_uaps --in-chns 1 --classes 2 --aux 5 --batch 8 --size 512 --steps 6 --warmup 2 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 0 $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs3 $1', d['ms_per_step'], d['config'].get('final_loss'))
"; }
for rep in 1 2 3 4; do
  UAPS_LAZY_BN_BWD=0 run "strips, eager, streams, one-piece BN" "--no-graph"
  run "strips, eager, streams, two halves" "--no-graph"
done 2>&1 | tee gpurun_out/r4z5_det.txt
