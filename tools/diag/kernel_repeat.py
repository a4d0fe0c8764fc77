#!/usr/bin/env python3
"""Single-kernel bit repeatability: each conv direction of a list of layer shapes is launched REPS times on fixed inputs and
every result is compared bit for bit with the first one.  Start two of these at once to time-share the card:
  python tools/diag/kernel_repeat.py MODE REPS [tag]
Prints one line per (direction, shape) with the number of launches whose output differed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uaps_amd import conv as C, bounds

mode, reps = sys.argv[1], int(sys.argv[2])
tag = sys.argv[3] if len(sys.argv) > 3 else "a"
dev = torch.device("cuda:0")
C.set_mode(mode)
torch.manual_seed(11)
# (B, Cin, Cout, HW): the small net of tests/test_gpu_two_ranks.py (4 images) and the bench net (32 images)
SHAPES = [(4, 16, 16, 32), (4, 32, 32, 16), (4, 64, 64, 8), (4, 128, 128, 4), (4, 32, 16, 32), (4, 16, 8, 64),
          (32, 32, 32, 128), (32, 64, 64, 64), (32, 128, 128, 32), (32, 16, 16, 256), (32, 32, 16, 256), (32, 64, 32, 128)]
if os.environ.get("KR_SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in os.environ["KR_SHAPES"].split(";")]
t0 = time.time()
for (B, Cin, Cout, HW) in SHAPES:
    x = torch.randn(B, Cin, HW, HW, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    dy = torch.randn(B, Cout, HW, HW, device=dev)
    wf, wb = C.pack_weights(w)
    xb = (bounds.from_value(x.abs().max()), 1.0) if mode == "h16" else None
    dyb = (bounds.from_value(dy.abs().max()), 1.0) if mode == "h16" else None
    for d in ("fwd", "fwd_stats", "bwd", "wrw"):
        def run():
            if d == "fwd":
                return (C.conv_fwd_raw(x, wf, None, Cout, 3, 0, xb=xb),)
            if d == "fwd_stats":
                y, st, _ = C.conv_fwd_raw(x, wf, None, Cout, 3, 0, want_stats=True, xb=xb)
                return (y, st)
            if d == "bwd":
                return (C.conv_bwd_data_raw(dy, wb, Cin, 3, 0, dyb=dyb),)
            dw, db = C.conv_bwd_weight_raw(dy, x, 3, True, 0, dyb=dyb, xb=xb)
            return (dw, db)
        ref = [t.clone() for t in run()]
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        maxd = torch.zeros((), dtype=torch.float32, device=dev)
        for _ in range(reps):
            out = run()
            ne = torch.zeros((), dtype=torch.bool, device=dev)
            for o, r in zip(out, ref):
                ne = ne | (o.view(torch.int32) != r.view(torch.int32)).any()
                maxd = torch.maximum(maxd, (o - r).abs().max())
            bad += ne
        torch.cuda.synchronize()
        print(f"[{tag} {mode}] {d:9s} B={B} {Cin}->{Cout} @{HW}: {int(bad)} of {reps} launches differ, max |diff| {float(maxd):.3e}  ({time.time() - t0:.0f}s)", flush=True)
