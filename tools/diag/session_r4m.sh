#!/bin/bash
export TMPDIR=/tmp
python tools/diag/aten_kernels.py --size 96 --batch 2 --stacks > gpurun_out/r4m_aten_small.txt 2>&1
python tools/diag/aten_kernels.py --size 640 --batch 8 > gpurun_out/r4m_aten_640.txt 2>&1
tail -30 gpurun_out/r4m_aten_640.txt
