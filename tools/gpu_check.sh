#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, smoke, bench, rocprof kernel stats.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing|gfx" | head -4
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log; tail -15 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
