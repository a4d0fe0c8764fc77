#!/bin/bash
export TMPDIR=/tmp
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 4 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['single_stream']['ms_per_step'], d['config']['final_loss'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['calls_per_step'])[:60]:
    if 'wrw' in k: print('   %7.1f' % (v['avg_us']*v['calls_per_step']), k, v)
"; }
unset UAPS_LAZY_BN_BWD; run "two halves"
export UAPS_LAZY_BN_BWD=0; run "one piece"
