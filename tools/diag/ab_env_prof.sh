#!/bin/bash
# same-box A/B of one environment switch, per kernel: two single-stream eager rocprofv3 kernel-trace runs of the bench step
# (arm A: the variable set, arm B: default), then tools/diag/prof_diff.py prints the kernels whose time per step differs.
#   tools/diag/ab_env_prof.sh VAR=VALUE [name-filter]
KV=$1; FILT=${2:-}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for arm in A B; do
  if [ $arm = A ]; then export $KV; else unset ${KV%%=*}; fi
  rm -rf $R/gpurun_out/abprof_$arm
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abprof_$arm -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --analysis-steps 0 --exact-steps 0 --single-stream --no-graph --no-cpu-baseline --no-inference --other-configs 0 > $R/gpurun_out/abprof_$arm.log 2>&1
  find $R/gpurun_out/abprof_$arm -name "*kernel_trace.csv" -delete
done
cd $R
python3 tools/diag/prof_diff.py $(find gpurun_out/abprof_A -name "*kernel_stats.csv" | head -1) $(find gpurun_out/abprof_B -name "*kernel_stats.csv" | head -1) 13 "$FILT"
