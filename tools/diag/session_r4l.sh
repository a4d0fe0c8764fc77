#!/bin/bash
# round 4, session l: ResNet joins with handles
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_res_uaps.py tests/test_gpu_strided.py tests/test_gpu_fused_ops.py -x -q -m gpu > gpurun_out/r4l_tests.txt 2>&1
tail -6 gpurun_out/r4l_tests.txt
python bench.py --net resnet50_uaps --size 640 --classes 2 --batch 8 --steps 6 --warmup 3 --no-cpu-baseline --other-configs 0 --exact-steps 0 --analysis-steps 0 2>gpurun_out/r4l_bench.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resnet50 640', d['value'], d['ms_per_step'])
" | tee gpurun_out/r4l_bench.txt
