#!/bin/bash
# 1x1 GEMM kernels: 256 output channels per workgroup (default where 256 divide the width) against 128 (UAPS_DIAG_G1_NARROW=1). GPU box.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
out=$R/gpurun_out/g1_wide_ab.txt
: > $out
for rep in 1 2; do
  echo "== default (256-wide blocks where they divide), repetition $rep" >> $out
  timeout 300 python3 $R/tools/bench_1x1.py 2>&1 | grep -v amdgpu.ids >> $out
  echo "== UAPS_DIAG_G1_NARROW=1, repetition $rep" >> $out
  UAPS_DIAG_G1_NARROW=1 timeout 300 python3 $R/tools/bench_1x1.py 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
