#!/bin/bash
# Round 6: same-box interleaved A/B of the whole-layer-width tile kernel (csrc/conv_split_g.hpp) on the headline step:
# arm T = UAPS_DIAG_NO_G=1 (the round-5 8 x 32-tile kernels), arm G = default (conv_hg128_kernel), arm D = UAPS_DIAG_G_DEEP=1 (all nine
# taps' weight fragments in registers, loaded around the fetch).   bash tools/diag/ab_g128.sh [reps]
N=${1:-4}
for rep in $(seq $N); do
  for arm in T G D; do
    unset UAPS_DIAG_NO_G UAPS_DIAG_G_DEEP
    [ $arm = T ] && export UAPS_DIAG_NO_G=1
    [ $arm = D ] && export UAPS_DIAG_G_DEEP=1
    timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --analysis-steps 0 --exact-steps 0 --other-configs 0 --no-inference --no-power-log 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', d['ms_per_step'], 'ms/step', d['value'], 'img/s', d['step_ms']['p50'], d['config'].get('switches'))"
  done
done
