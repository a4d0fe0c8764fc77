#!/bin/bash
# Memory-side traffic and duration of one conv launch configuration: tools/diag/layer_traffic.sh tag Cin Cout HW ks cfg mode [dir] [B]
# (separate --pmc passes for FETCH_SIZE, WRITE_SIZE and the raw L2 request counters; a kernel-trace pass for the duration)
export TMPDIR=/tmp
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/ltraf/$tag
mkdir -p $out
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/p1 -o pmc -- python3 $GRAFT_REPO_ROOT/tools/diag/layer_run.py "$@" > $out/p1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/p2 -o pmc -- python3 $GRAFT_REPO_ROOT/tools/diag/layer_run.py "$@" > $out/p2.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/p3 -o pmc -- python3 $GRAFT_REPO_ROOT/tools/diag/layer_run.py "$@" > $out/p3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/p4 -o kt -- python3 $GRAFT_REPO_ROOT/tools/diag/layer_run.py "$@" > $out/p4.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$out" "$tag" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "uaps::conv" not in k or "pack" in k or "reduce" in k: continue
        a = acc[k[:60]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Name"]
        if "uaps::conv" not in k or "pack" in k or "reduce" in k: continue
        acc[k[:60]]["avg_us"] = [float(r["AverageNs"]) / 1e3, 1]
for k, d in acc.items():
    print(sys.argv[2], k)
    for c, (v, n) in sorted(d.items()):
        print(f"    {c:28s} {v / n:16.1f}")
PY
find $out -name "*counter_collection.csv" -size +2M -delete
find $out -name "*kernel_trace.csv" -delete
