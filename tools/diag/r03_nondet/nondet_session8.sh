#!/bin/bash
# the final build of round 3: the deviating pair and the traced steps again (both stream settings, modes 1 and 2)
mkdir -p gpurun_out/nd8
O=gpurun_out/nd8
export TMPDIR=/tmp
R=${R:-40000}
export UAPS_CONV_MODE=1
( timeout 900 python tools/diag/outconv_repeat.py $R 1 > $O/x.log 2>&1 ) &
D=$!
timeout 900 python tools/diag/outconv_repeat.py $R 1 > $O/y.log 2>&1
wait $D
grep -h pid $O/x.log $O/y.log
for mode in 2 1; do
  for streams in 0 1; do
    export UAPS_CONV_MODE=$mode UAPS_TEST_STREAMS=$streams
    timeout 300 python tools/diag/share_repeat.py ref /tmp/ref_${mode}_${streams}.pt > $O/ref_${mode}_${streams}.log 2>&1
    ( timeout 900 python tools/diag/share_repeat.py check 40 /tmp/ref_${mode}_${streams}.pt > $O/chk_${mode}_${streams}_a.log 2>&1 ) &
    A=$!
    ( timeout 900 python tools/diag/share_repeat.py check 40 /tmp/ref_${mode}_${streams}.pt > $O/chk_${mode}_${streams}_b.log 2>&1 ) &
    B=$!
    wait $A $B
    echo "mode $mode streams $streams:"; tail -1 $O/chk_${mode}_${streams}_a.log; tail -1 $O/chk_${mode}_${streams}_b.log
  done
done
unset UAPS_CONV_MODE UAPS_TEST_STREAMS
