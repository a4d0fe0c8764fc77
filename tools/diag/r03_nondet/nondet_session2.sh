#!/bin/bash
# Round 3, nondeterminism hunt, session 2: trace the first deviating launch (single stream, two processes sharing the card).
mkdir -p gpurun_out/nd2
O=gpurun_out/nd2
export TMPDIR=/tmp UAPS_TEST_STREAMS=0
N=${N:-40}
for mode in 2 1 0; do
  export UAPS_CONV_MODE=$mode
  timeout 300 python tools/diag/trace_repeat.py ref /tmp/tref_${mode}.pt > $O/ref_${mode}.log 2>&1
  ( timeout 900 python tools/diag/trace_repeat.py check $N /tmp/tref_${mode}.pt > $O/chk_${mode}_a.log 2>&1 ) &
  A=$!
  ( timeout 900 python tools/diag/trace_repeat.py check $N /tmp/tref_${mode}.pt > $O/chk_${mode}_b.log 2>&1 ) &
  B=$!
  wait $A $B
  echo "== mode $mode"; tail -2 $O/ref_${mode}.log; grep -h "pid" $O/chk_${mode}_a.log $O/chk_${mode}_b.log | cut -c1-600
done
