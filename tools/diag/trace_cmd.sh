#!/bin/bash
# kernel-trace stats of a python command: tools/diag/trace_cmd.sh tag script.py [args]
export TMPDIR=/tmp
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/trace/$tag
mkdir -p $out
script=$GRAFT_REPO_ROOT/$1; shift
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $script "$@" > $out/run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f}")
PY
find $out -name "*kernel_trace.csv" -size +4M -delete
