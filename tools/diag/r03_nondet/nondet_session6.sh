#!/bin/bash
mkdir -p gpurun_out/nd6
O=gpurun_out/nd6
export TMPDIR=/tmp
P=tools/bin/pkfma_probe
L=${L:-1500}
for kind in 0 1 2 3; do
  echo "--- beside a hammer process of kind $kind"
  ( timeout 200 $P hammer 100 $kind > $O/hammer$kind.log 2>&1 ) &
  H=$!
  sleep 2
  for f in 2 4 5 6 7 3 0; do timeout 300 $P probe $f $L | head -2 | tee -a $O/kind$kind.log; done
  kill $H 2>/dev/null; wait $H 2>/dev/null
done
