#!/bin/bash
# HBM traffic of single GEMM-tiled 1x1 layers (GPU box): two --pmc passes per shape, no trace domains
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for shape in "256 1024 80" "1024 256 80" "2048 512 80" "512 2048 80"; do
  tag=$(echo $shape | tr ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/g1pmc/$tag/$c -o pmc -- python3 $R/tools/diag/g1_traffic.py $shape > $R/gpurun_out/g1pmc/$tag.$c.log 2>&1
  done
  python3 - $R/gpurun_out/g1pmc/$tag "$shape" <<'PY'
import csv, glob, sys, os
root, shape = sys.argv[1], sys.argv[2]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c and "conv_g1h" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    out[c] = vals
f = [v * 1024 / 1e6 for v in out["FETCH_SIZE"]]; w = [v * 1024 / 1e6 for v in out["WRITE_SIZE"]]
print(shape, "FETCH raw MB per launch", [round(v, 1) for v in f], "WRITE MB", [round(v, 1) for v in w])
PY
  grep algorithmic $R/gpurun_out/g1pmc/$tag.FETCH_SIZE.log
done 2>&1 | tee $R/gpurun_out/r4v_g1_traffic.txt
find $R/gpurun_out/g1pmc -name "*.csv" -size +1M -delete
