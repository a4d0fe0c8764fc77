#!/bin/bash
export TMPDIR=/tmp
python bench.py --net resnet50_uaps --size 640 --classes 2 --batch 8 --steps 6 --warmup 3 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resnet50 640', d['value'], d['ms_per_step'])
tot=0
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_us']*kv[1]['calls_per_step'])[:40]:
    print('   %8.1f us/step' % (v['avg_us']*v['calls_per_step']), k, v)
" | tee gpurun_out/r4q_resnet.txt
