#!/bin/bash
# A/B of the loss-block kernels: per-kernel average durations (rocprofv3 --kernel-trace --stats) of tools/bench_loss.py for
# each library given as LIBS="name=path ..." (default: the in-tree one).   bash tools/diag/loss_ab.sh [cfg]
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
CFG=${1:-0}
LIBS=${LIBS:-"tree=uaps_amd/lib/libuaps_hip.so"}
mkdir -p gpurun_out
for kv in $LIBS; do
  name=${kv%%=*}; path=${kv#*=}
  export UAPS_HIP_LIB=$PWD/$path
  out=gpurun_out/loss_ab_$name
  rm -rf $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 tools/bench_loss.py --cfg $CFG > $out.log 2>&1
  echo "== $name cfg=$CFG"; tail -2 $out.log
  f=$(find $out -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'pair_' in n or 'finalize' in n:
        print(f"   {n[:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
done
