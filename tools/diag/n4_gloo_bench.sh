#!/bin/bash
# Round 5 (review item 9): the N > 1 code path of bench.py on ONE card -- 4 ranks over gloo (RCCL refuses two ranks on a device), the whole job
# confined to 8 host cores = 2 cores per rank -- eager step against the two-graph step (uaps_amd/graph.py), at the metric's batch and at a
# host-bound one.  No scaling figure comes out of this (four ranks share one GPU); it shows what the launch path costs a core-starved rank.
export UAPS_BENCH_BACKEND=gloo UAPS_BENCH_DEVICE=0
for batch in 16 4; do
  for gm in 0 1 auto; do
    if [ $gm = auto ]; then unset UAPS_GRAPH_MULTI; else export UAPS_GRAPH_MULTI=$gm; fi
    taskset -c 0-7 timeout 900 python bench.py --gpus 4 --batch $batch --steps 10 --warmup 4 --no-cpu-baseline --analysis-steps 0 --exact-steps 0 --other-configs 0 --no-inference 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $batch+$batch per rank, UAPS_GRAPH_MULTI=$gm:', d['ms_per_step'], 'ms/step', d['value'], 'img/s (4 ranks, one card)', 'ranks_seen', d['ranks_seen'], 'cores/rank', d['config'].get('cores_per_rank'), '|', d['config']['launch_mode'][:60])"
  done
done
