#!/bin/bash
# HBM traffic counters of another configuration of the bench command (GPU box): tools/gpu_pmc.sh with a tag and bench arguments,
#   bash tools/gpu_pmc_cfg.sh configs3 --net unet_uaps --in-chns 1 --classes 2 --aux 5 --batch 8 --size 512
# -> gpurun_out/pmc_<tag>/{summary.txt,pmc_traffic.json}; copy the json to profiles/pmc_traffic_<tag>.json (bench.py other_configs).
tag=$1; shift
export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $D
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $D/$c -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --steps 2 --warmup 1 --analysis-steps 0 --exact-steps 0 --single-stream --no-graph --no-cpu-baseline --no-inference --other-configs 0 > $D/$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py gpurun_out/pmc_$tag 3 > $D/summary.txt 2>&1
head -12 $D/summary.txt
find $D -name "*counter_collection.csv" -size +8M -delete
