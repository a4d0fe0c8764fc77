#!/bin/bash
# same-box A/B of TWO TREES (e.g. the previous round's, unpacked and built under _r04ab/) on the headline step, interleaved:
#   tools/diag/ab_tree_bench.sh _r04ab . [reps]        (bench.py of each tree runs from inside that tree)
A=$1; B=$2; N=${3:-3}
R=$PWD
for rep in $(seq $N); do
  for t in $A $B; do
    extra=""; grep -q "no-inference" $R/$t/bench.py && extra="--no-inference"
    (cd $R/$t && timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --analysis-steps 0 --exact-steps 0 --other-configs 0 $extra 2>/dev/null) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tree $t:', d['ms_per_step'], 'ms/step', d['value'], 'img/s', 'p50', d['step_ms']['p50'])"
  done
done
