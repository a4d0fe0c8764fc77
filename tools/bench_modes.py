#!/usr/bin/env python3
"""Per-layer timing of the convolution kernels in both arithmetic modes (split-bf16 vs exact fp32 MFMA) at the layer shapes
of the bench step (B = 32), with the split kernel variants forced through cfg bits 29-30 (1 = 16x16x32 form, 2 = 32x32x16
form; low byte = output channels per workgroup).  Run on the GPU box: python tools/bench_modes.py [--only substr]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uaps_amd import conv as C
from tools.bench_conv import LAYERS, timeit

VARIANTS = [("exact", "exact", 0), ("auto", "split", 0), ("s16", "split", 1 << 29), ("s32/32", "split", (2 << 29) | 32), ("s32/64", "split", (2 << 29) | 64)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--dirs", type=str, default="fwd,bwd")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.batch
    dirs = args.dirs.split(",")
    tot = {}
    print(f"{'layer':26s} {'GF':>6s} | " + " | ".join(" ".join(f"{d[0]}:{v[0]:>7s}" for v in VARIANTS) for d in dirs) + "   (us; TF/s of auto)")
    for name, Cin, Cout, HW, ks, calls in LAYERS:
        if args.only and args.only not in name:
            continue
        x = torch.randn(B, Cin, HW, HW, device=dev)
        w = torch.randn(Cout, Cin, ks, ks, device=dev) * 0.05
        dy = torch.randn(B, Cout, HW, HW, device=dev)
        wf, wb = C.pack_weights(w)
        gf = 2.0 * B * HW * HW * Cin * Cout * ks * ks / 1e9
        cols = []
        for d in dirs:
            row = []
            for vname, mode, cfg in VARIANTS:
                C.set_mode(mode)
                if d == "wrw":
                    if vname in ("s16", "s32/32", "s32/64"):
                        row.append(float("nan"))
                        continue
                    wcfg = 0          # the mode decides (exact kernels in exact mode, split where the plan takes them)
                    fn = lambda: C.conv_bwd_weight_raw(dy, x, ks, False, wcfg)
                else:
                    fn = (lambda: C.conv_fwd_raw(x, wf, None, Cout, ks, cfg)) if d == "fwd" else (lambda: C.conv_bwd_data_raw(dy, wb, Cin, ks, cfg))
                try:
                    t = timeit(fn, iters=20)
                except Exception:
                    t = float("nan")
                row.append(t)
                if vname in ("exact", "auto"):
                    tot[(d, vname)] = tot.get((d, vname), 0.0) + t * calls
            cols.append(" ".join(f"{t:9.1f}" for t in row) + f" {gf / row[1] * 1e3:6.1f}")
        print(f"{name:26s} {gf:6.2f} | " + " | ".join(cols), flush=True)
    C.set_mode("split")
    print("sum over the 4-head net, one pass of the batch (us): " + ", ".join(f"{d}/{m}={v:.0f}" for (d, m), v in tot.items()))


if __name__ == "__main__":
    main()
