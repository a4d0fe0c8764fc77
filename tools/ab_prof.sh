set -x
export TMPDIR=/tmp
cd /tmp
for v in 0 1; do
UAPS_EPILOGUE_STATS=$v rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ab$v -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_ab$v.log 2>&1
done
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_ab0 gpurun_out/prof_ab1 -name "*kernel_trace.csv" -delete
