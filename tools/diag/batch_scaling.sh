#!/bin/bash
# Does a 4x larger launch (what one grouped launch over the four decoders would be) run a layer's images faster than four launches?
export TMPDIR=/tmp
for b in 32 128 32 128; do
  echo "=== batch $b"
  python tools/bench_conv.py --batch $b --no-miopen --only dec. 2>/dev/null
done > gpurun_out/batch_scaling.txt 2>&1
