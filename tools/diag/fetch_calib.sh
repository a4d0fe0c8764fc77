#!/bin/bash
# FETCH_SIZE calibration (tools/diag/fetch_calib.hip): three separate --pmc passes, then per-kernel averages.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/calib $R/tools/bin
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/diag/fetch_calib.hip -o $R/tools/bin/fetch_calib || exit 1
cd /tmp
for c in FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; do
  rm -rf $R/gpurun_out/calib/$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/calib/$c -o pmc -- $R/tools/bin/fetch_calib > $R/gpurun_out/calib/$c.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"gpurun_out/calib/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                a = acc[r["Kernel_Name"].split("(")[0]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in acc.items():
        res[k][c] = v / max(n, 1)
print(open("gpurun_out/calib/FETCH_SIZE.log").read().strip().splitlines()[-1])
print(f"{'kernel':40s} {'FETCH_SIZE (KiB -> B)':>22s} {'RDREQ':>14s} {'RDREQ_32B':>12s} {'2*F - 32*R32 (B)':>18s}")
for k, d in sorted(res.items()):
    f = d.get("FETCH_SIZE", 0) * 1024; r = d.get("TCC_EA0_RDREQ_sum", 0); r32 = d.get("TCC_EA0_RDREQ_32B_sum", 0)
    print(f"{k[:40]:40s} {f:22.0f} {r:14.0f} {r32:12.0f} {2 * f - 32 * r32:18.0f}")
PY
