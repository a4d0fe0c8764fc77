#!/bin/bash
mkdir -p gpurun_out/nd5
O=gpurun_out/nd5
export TMPDIR=/tmp
P=tools/bin/pkfma_probe
L=${L:-4000}
echo "--- alone"; for f in 1 2; do timeout 300 $P probe $f $L | tee -a $O/alone.log; done
echo "--- beside a hammer PROCESS"
( timeout 200 $P hammer 100 > $O/hammer.log 2>&1 ) &
H=$!
sleep 2
for f in 0 1 2 3; do timeout 300 $P probe $f $L | tee -a $O/beside.log; done
kill $H 2>/dev/null; wait $H 2>/dev/null
echo "--- hammer on a second stream of the same process"
for f in 0 1 2 3; do timeout 300 $P probe $f $L both | tee -a $O/both.log; done
