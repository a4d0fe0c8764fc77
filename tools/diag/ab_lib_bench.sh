#!/bin/bash
# same-box A/B of two builds of libuaps_hip.so on the headline step, interleaved: tools/diag/ab_lib_bench.sh base.so new.so [reps]
A=$1; B=$2; N=${3:-3}
for rep in $(seq $N); do
  for lib in $A $B; do
    UAPS_HIP_LIB=$PWD/$lib timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --analysis-steps 0 --exact-steps 0 --other-configs 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
  done
done
