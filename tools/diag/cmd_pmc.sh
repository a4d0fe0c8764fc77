#!/bin/bash
# SQ counters of the kernels matching a name filter for any python command: tools/diag/cmd_pmc.sh tag filter script.py [args...]
export TMPDIR=/tmp
tag=$1; filt=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/lpmc/$tag
mkdir -p $out
cd /tmp
script=$GRAFT_REPO_ROOT/$1; shift
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/p1 -o pmc -- python3 $script "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $out/p2 -o pmc -- python3 $script "$@" > $out/p2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/p3 -o pmc -- python3 $script "$@" > $out/p3.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$out" "$filt" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sys.argv[2] not in k: continue
        k = k[:70]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in acc.items():
    ds = sorted(dur[k])
    print(k, "median duration ns", ds[len(ds) // 2])
    for c, (v, n) in sorted(d.items()):
        print(f"    {c:28s} {v / n:16.0f}  (per launch, {n} launches)")
PY
find $out -name "*counter_collection.csv" -size +2M -delete
