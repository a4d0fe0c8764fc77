#!/bin/bash
# session 3: (a) the deviating launch pair alone, on fixed inputs, beside a disturbing process; (b) traces with the forwards' saved tensors
mkdir -p gpurun_out/nd3
O=gpurun_out/nd3
export TMPDIR=/tmp UAPS_TEST_STREAMS=0
R=${R:-30000}
export UAPS_CONV_MODE=1
timeout 300 python tools/diag/trace_repeat.py ref /tmp/tref.pt > $O/ref.log 2>&1
# (a) pair (ctx 0 and 1) beside a trace_repeat in mode 1
( timeout 900 python tools/diag/trace_repeat.py check 60 /tmp/tref.pt > $O/a_disturb.log 2>&1 ) &
D=$!
timeout 900 python tools/diag/outconv_repeat.py $R 0 > $O/a_pair0.log 2>&1
timeout 900 python tools/diag/outconv_repeat.py $R 1 > $O/a_pair1.log 2>&1
wait $D
tail -1 $O/a_pair0.log; tail -1 $O/a_pair1.log; grep -h pid $O/a_disturb.log | cut -c1-400 | tail -8
# (a2) two pair processes beside each other
( timeout 900 python tools/diag/outconv_repeat.py $R 1 > $O/a2_x.log 2>&1 ) &
D=$!
timeout 900 python tools/diag/outconv_repeat.py $R 1 > $O/a2_y.log 2>&1
wait $D
tail -1 $O/a2_x.log; tail -1 $O/a2_y.log
# (b) traces with saved tensors, two processes
( timeout 900 python tools/diag/trace_repeat.py check 40 /tmp/tref.pt > $O/b_a.log 2>&1 ) &
A=$!
( timeout 900 python tools/diag/trace_repeat.py check 40 /tmp/tref.pt > $O/b_b.log 2>&1 ) &
B=$!
wait $A $B
grep -h pid $O/b_a.log $O/b_b.log | cut -c1-500
