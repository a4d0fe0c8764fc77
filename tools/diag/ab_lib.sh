#!/bin/bash
# A/B of library builds tools/bin/libuaps_<tag>.so on one box, interleaved: tools/diag/ab_lib.sh "<bench_conv --only list>" tag1 tag2 ...
only=$1; shift
mkdir -p gpurun_out/abl
for rep in 1 2; do
 for v in "$@"; do
  UAPS_HIP_LIB=$PWD/tools/bin/libuaps_$v.so timeout 300 python tools/bench_conv.py --batch 32 --mode h16 --only "$only" > gpurun_out/abl/conv_${v}_$rep.txt 2>&1
 done
done
for v in "$@"; do echo "== $v"; grep -E "@" gpurun_out/abl/conv_${v}_2.txt | cut -c1-130; done
