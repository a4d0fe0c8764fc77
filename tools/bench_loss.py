#!/usr/bin/env python3
"""Times the one-launch loss block (uaps_pairloss_fwd / _bwd) at the BASELINE.json sizes, per launch with HIP events, and
prints GB/s on the algorithmic bytes (forward 4DC + 8 per pixel and branch, backward 8DC + 8).
  python tools/bench_loss.py [--cfg 0,256,512,1024]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import uaps_amd
from uaps_amd import losses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="0")
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for D, B, C, H, W in [(4, 16, 4, 256, 256), (6, 8, 2, 512, 512)]:
        both = [torch.randn(2 * B, C, H, W, device=dev) * 2 for _ in range(D)]
        for t in both:
            t.requires_grad_(True)
        y = torch.randint(0, C, (B, H, W), device=dev)
        w = np.random.default_rng(0).dirichlet(np.ones(D))
        N = B * H * W
        fb, bb = N * (8 * D * C + 16), N * 2 * (8 * D * C + 8)
        for cfg in [int(c, 0) for c in args.cfg.split(",")]:
            losses.PAIR_CFG = cfg
            tf, tb = [], []
            for it in range(args.iters + 5):
                losses.KERNEL_EVENTS = {}
                out = uaps_amd.uaps_pair_loss(both, y, w, 0.1, 0.1)
                out.loss.backward()
                torch.cuda.synchronize()
                ev = losses.KERNEL_EVENTS
                if it >= 5:
                    tf.append(ev["uaps_pair_fwd"][0][0].elapsed_time(ev["uaps_pair_fwd"][0][1]) * 1e3)
                    tb.append(ev["uaps_pair_bwd"][0][0].elapsed_time(ev["uaps_pair_bwd"][0][1]) * 1e3)
            losses.KERNEL_EVENTS = None
            f, b = float(np.median(tf)), float(np.median(tb))
            print(f"D={D} C={C} B={B} {H}x{W} cfg={cfg}: fwd(+finalize) {f:7.1f} us {fb / f / 1e3:7.1f} GB/s | bwd {b:7.1f} us {bb / b / 1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
