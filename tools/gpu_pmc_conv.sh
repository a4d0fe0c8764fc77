#!/bin/bash
# SQ stall / LDS counters of the conv kernels for one layer of tools/bench_conv.py:  tools/gpu_pmc_conv.sh "dec.up2b"
set -x
export TMPDIR=/tmp
layer=${1:-dec.up2b}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv
mkdir -p $out
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/p1 -o pmc -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --batch 32 --only "$layer" > $out/p1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $out/p2 -o pmc -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --batch 32 --only "$layer" > $out/p2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/pmc_conv/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "uaps::conv" not in k: continue
        a = acc[k[:70]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(k)
    for c, (v, n) in sorted(d.items()):
        print(f"    {c:28s} {v / n:16.0f}  (per launch, {n} launches)")
PY
find gpurun_out/pmc_conv -name "*counter_collection.csv" -size +4M -delete
