#!/bin/bash
# A/B of one environment switch on the SAME GPU box, interleaved (different boxes differ by several % in clocks):
#   tools/ab_bench.sh UAPS_EPILOGUE_STATS 0 1
var=$1; a=$2; b=$3
mkdir -p gpurun_out
for rep in 1 2 3; do
  for v in $a $b; do
    env $var=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$var=$v', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
  done
done
