#!/usr/bin/env python3
"""MFMA shape A/B at the same per-wave output tile (guide rule 28): the 16x16x32 form (conv_hfwd_kernel<3,8,32,32,16>, cfg bit 29)
against the 32x32x16 form (conv_h32_kernel<32|64>, cfg bits 30 | BN) and the plan's own choice, fp16-split arithmetic with bounds,
random data, back-to-back launches.  python tools/diag/shape_ab.py [--batch 32]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uaps_amd import conv as C, bounds
from tools.bench_conv import timeit

LAYERS = [("32->32@128", 32, 32, 128), ("64->32@128", 64, 32, 128), ("64->64@64", 64, 64, 64), ("128->64@64", 128, 64, 64),
          ("128->128@32", 128, 128, 32), ("256->128@32", 256, 128, 32), ("256->256@16", 256, 256, 16), ("128->256@16", 128, 256, 16), ("256->128@16", 256, 128, 16), ("128->256@32", 128, 256, 32), ("16->32@128", 16, 32, 128)]
VARIANTS = [("auto", 0), ("16x16x32/32", 1 << 29), ("32x32x16/32", (2 << 29) | 32), ("32x32x16/64", (2 << 29) | 64)]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    C.set_mode("h16")
    B = args.batch
    print(f"{'layer':14s} " + " ".join(f"{v[0]:>14s}" for v in VARIANTS) + "   (us, min of reps; fwd)")
    for name, Cin, Cout, HW in LAYERS:
        x = torch.randn(B, Cin, HW, HW, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
        wf, wb = C.pack_weights(w)
        xb = (bounds.from_value(x.abs().max()), 1.0)
        best = {}
        for rep in range(args.reps):
            for vname, cfg in VARIANTS:
                try:
                    t = timeit(lambda: C.conv_fwd_raw(x, wf, None, Cout, 3, cfg, xb=xb), iters=30, warm=5)
                except Exception as ex:
                    t = float("nan")
                best[vname] = min(best.get(vname, 1e9), t)
        gf = 2.0 * B * HW * HW * Cin * Cout * 9 / 1e9
        print(f"{name:14s} " + " ".join(f"{best[v[0]]:8.1f}/{gf / best[v[0]] * 1e3:5.0f}" for v in VARIANTS), flush=True)

if __name__ == "__main__":
    main()
