#!/bin/bash
# memory-path counters of one conv launch configuration: tools/diag/layer_pmc_mem.sh tag Cin Cout HW ks cfg mode [dir] [B]
export TMPDIR=/tmp
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/lpmc/$tag
mkdir -p $out
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TD|TCC|SQC)_[A-Z0-9_]+" | sort -u > $out/counters.txt
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $out/m1 -o pmc -- python3 $GRAFT_REPO_ROOT/tools/diag/layer_run.py "$@" > $out/m1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $out/m2 -o pmc -- python3 $GRAFT_REPO_ROOT/tools/diag/layer_run.py "$@" > $out/m2.log 2>&1
rocprofv3 --pmc TD_TD_BUSY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum --output-format csv -d $out/m3 -o pmc -- python3 $GRAFT_REPO_ROOT/tools/diag/layer_run.py "$@" > $out/m3.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$out" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/m*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "uaps::conv" not in k or "pack" in k: continue
        a = acc[k[:60]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(k)
    for c, (v, n) in sorted(d.items()):
        print(f"    {c:32s} {v / n:16.0f}  (per launch, {n} launches)")
PY
tail -3 $out/m1.log $out/m3.log | grep -i "error\|invalid" | head
find $out -name "*counter_collection.csv" -size +2M -delete
