#!/bin/bash
# 1x1 GEMM kernels (conv_g1_body): the shipped build against the ablations of `make -C uaps_amd/csrc g1abl` (GPU box).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
out=$R/gpurun_out/g1_ablate.txt
: > $out
for v in "" ${G1_VARIANTS:-_g1e1 _g1e2 _g1e3 _g1e4 _g1e5 _g1e6 _g1e7}; do
  echo "== libuaps_hip$v.so" >> $out
  UAPS_HIP_LIB=$R/uaps_amd/lib/libuaps_hip$v.so timeout 300 python3 $R/tools/bench_1x1.py >> $out 2>&1
done
cat $out
