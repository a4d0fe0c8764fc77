#!/bin/bash
# Same-box ablation of the round-2 mechanisms: one bench.py run per configuration (GPU box), images/s and ms/step.
# usage: bash tools/ablation.sh > gpurun_out/ablation.txt
run() {   # label, env..., -- bench args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  out=$(env "${envs[@]}" python bench.py --no-cpu-baseline --analysis-steps 0 --exact-steps 0 "$@" 2>/dev/null | tail -1)
  python - "$label" "$out" <<'PY'
import json, sys
r = json.loads(sys.argv[2])
print(f"{sys.argv[1]:78s} {r['value']:8.1f} img/s  {r['ms_per_step']:7.3f} ms/step")
PY
}
run "default: fp16-split convs, hipGraph replay, decoder streams" X=1 --
run "eager launches instead of the hipGraph replay (--no-graph)" X=1 -- --no-graph
run "single stream (--single-stream), hipGraph" X=1 -- --single-stream
run "single stream, eager" X=1 -- --single-stream --no-graph
run "UAPS_CONV_MODE=1: three-piece bf16 split (six products), graph + streams" UAPS_CONV_MODE=1 --
run "UAPS_CONV_MODE=0: fp32 matrix instruction everywhere, graph + streams" UAPS_CONV_MODE=0 --
run "UAPS_SWRW_COLMAJOR=0: row-major weight-gradient tile order" UAPS_SWRW_COLMAJOR=0 --
run "UAPS_WRW_TALL=0: 4-row weight-gradient tiles at 256x256" UAPS_WRW_TALL=0 --
run "UAPS_STAT_SHIFT=0: BatchNorm partial sums without the running-mean shift" UAPS_STAT_SHIFT=0 --
run "UAPS_FUSED_BN_CONV=0: BatchNorm + LeakyReLU materialised instead of applied in the consuming conv" UAPS_FUSED_BN_CONV=0 --
run "4 + 4 images (--batch 4), graph + streams" X=1 -- --batch 4
run "4 + 4 images, eager" X=1 -- --batch 4 --no-graph
run "64 + 64 images (--batch 64)" X=1 -- --batch 64 --steps 8 --warmup 4
