#!/bin/bash
# Round 3, nondeterminism hunt, session 1 (run through gpurun).  Two questions:
#  E1: does ONE split-kernel launch on fixed inputs ever change its output bits while another process shares the card?
#  E2: do the 20-step runs deviate under sharing with ONE stream (UAPS_TEST_STREAMS=0) as they do with decoder streams?
mkdir -p gpurun_out/nd1
O=gpurun_out/nd1
export TMPDIR=/tmp
N=${N:-12}
# E1
( timeout 600 python tools/diag/kernel_repeat.py split 200 p0 > $O/e1_split.log 2>&1 ) &
P0=$!
( timeout 600 python tools/diag/kernel_repeat.py h16 200 p1 > $O/e1_h16.log 2>&1 ) &
P1=$!
wait $P0 $P1
grep -v " 0 of " $O/e1_split.log $O/e1_h16.log | tail -20
echo "E1 done: $(grep -c ' 0 of ' $O/e1_split.log) + $(grep -c ' 0 of ' $O/e1_h16.log) clean lines"
# E2
for mode in 1 2; do
  for streams in 0 1; do
    export UAPS_CONV_MODE=$mode UAPS_TEST_STREAMS=$streams
    timeout 300 python tools/diag/share_repeat.py ref /tmp/ref_${mode}_${streams}.pt > $O/e2_ref_${mode}_${streams}.log 2>&1
    ( timeout 600 python tools/diag/share_repeat.py check $N /tmp/ref_${mode}_${streams}.pt > $O/e2_chk_${mode}_${streams}_a.log 2>&1 ) &
    A=$!
    ( timeout 600 python tools/diag/share_repeat.py check $N /tmp/ref_${mode}_${streams}.pt > $O/e2_chk_${mode}_${streams}_b.log 2>&1 ) &
    B=$!
    wait $A $B
    echo "mode $mode streams $streams:"; tail -1 $O/e2_chk_${mode}_${streams}_a.log; tail -1 $O/e2_chk_${mode}_${streams}_b.log
  done
done
