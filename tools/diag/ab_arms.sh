#!/bin/bash
# Same-box interleaved comparison of the headline step under several environment settings ("arms"):
#   bash tools/diag/ab_arms.sh REPS "" "UAPS_DIAG_NO_WRW64=1" "UAPS_DIAG_WRW_B21=1" ...
# per run: headline ms/step, the single-stream analysis pass, and the per-kernel figures of the kernels matching $AB_KERNELS (regex).
N=$1; shift
for rep in $(seq $N); do
  for arm in "$@"; do
    ( [ -n "$arm" ] && export $arm
      timeout 900 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --exact-steps 0 --other-configs 0 --no-inference --no-power-log 2>/dev/null | python -c "
import json,sys,re,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pat=re.compile(os.environ.get('AB_KERNELS','hwrw|wrw_reduce'))
ks={k:(v.get('calls_per_step'),v.get('avg_us')) for k,v in d.get('kernels',{}).items() if pat.search(k)}
tot=sum(c*u for c,u in ks.values() if c and u)
print('%-24s' % ('$arm' or 'default'), d['ms_per_step'], 'ms/step | single-stream', d['single_stream']['ms_per_step'], '| loss', d['config'].get('final_loss'), '| matched kernels us/step %.1f' % tot, ks)" )
  done
done
