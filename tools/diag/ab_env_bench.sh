#!/bin/bash
# same-box A/B of one environment switch on the headline step, interleaved: tools/diag/ab_env_bench.sh VAR=VALUE [reps] [extra bench args]
# (arm A: the variable set, arm B: default)
KV=$1; N=${2:-3}; shift; shift
for rep in $(seq $N); do
  for arm in A B; do
    if [ $arm = A ]; then export $KV; else unset ${KV%%=*}; fi
    timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --analysis-steps 0 --exact-steps 0 --other-configs 0 --no-inference "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', '$KV' if '$arm' == 'A' else 'default', d['ms_per_step'], 'ms/step', d['value'], 'img/s', d['config'].get('final_loss'))"
  done
done
