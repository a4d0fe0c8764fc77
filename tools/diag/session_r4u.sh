#!/bin/bash
# round 4, session u: 2-deep row fetch in the row weight-gradient kernels
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "not float64" > gpurun_out/r4u_tests.txt 2>&1
tail -3 gpurun_out/r4u_tests.txt
python tools/bench_conv.py --batch 32 --no-miopen --only "@256" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4u_layers.txt
