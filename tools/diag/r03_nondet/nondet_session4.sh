#!/bin/bash
mkdir -p gpurun_out/nd4
O=gpurun_out/nd4
export TMPDIR=/tmp UAPS_CONV_MODE=1
R=${R:-40000}
( timeout 900 python tools/diag/outconv_repeat.py $R 1 > $O/x.log 2>&1 ) &
D=$!
timeout 900 python tools/diag/outconv_repeat.py $R 1 > $O/y.log 2>&1
wait $D
grep pid $O/x.log $O/y.log
