#!/bin/bash
# round 4, session a: concurrency experiments + batch scaling + the new tests
export TMPDIR=/tmp
bash tools/diag/ab_queues.sh > gpurun_out/ab_queues.log 2>&1
bash tools/diag/batch_scaling.sh
timeout 900 python -m pytest tests/test_gpu_corun.py tests/test_gpu_errors.py -x -q -m gpu > gpurun_out/r4a_tests.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_two_ranks.py -x -q -m gpu -k "without_a_launcher or one_line" >> gpurun_out/r4a_tests.txt 2>&1
tail -5 gpurun_out/r4a_tests.txt
cat gpurun_out/ab_queues.txt
