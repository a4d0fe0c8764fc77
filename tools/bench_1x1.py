#!/usr/bin/env python3
"""Timing of the wide 1x1 convolutions (ResNet-50 bottleneck shapes at the configs[4] per-GPU batch) in the three directions:
python tools/bench_1x1.py [--mode h16|split|exact] [--batch 16]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uaps_amd import conv as C, bounds

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="h16")
ap.add_argument("--batch", type=int, default=16)
args = ap.parse_args()
C.set_mode(args.mode)
dev = torch.device("cuda:0")
SHAPES = [(64, 256, 160), (256, 64, 160), (256, 128, 160), (128, 512, 80), (512, 128, 80), (512, 256, 80), (256, 1024, 80), (1024, 256, 80),
          (1024, 512, 80), (512, 2048, 80), (2048, 512, 80)]
bnd = (lambda t: (bounds.from_value(t.abs().max()), 1.0)) if args.mode == "h16" else (lambda t: None)


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print(f"mode {args.mode}, batch {args.batch}:   layer            GFLOP |  fwd us  TF/s | dgrad us  TF/s |  wrw us  TF/s")
for Cin, Cout, HW in SHAPES:
    B = args.batch
    x = torch.randn(B, Cin, HW, HW, device=dev)
    w = torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05
    dy = torch.randn(B, Cout, HW, HW, device=dev)
    wf, wb = C.pack_weights(w)
    xb, dyb = bnd(x), bnd(dy)
    gf = 2.0 * B * HW * HW * Cin * Cout / 1e9
    t1 = timeit(lambda: C.conv_fwd_raw(x, wf, None, Cout, 1, 0, xb=xb))
    t2 = timeit(lambda: C.conv_bwd_data_raw(dy, wb, Cin, 1, 0, dyb=dyb))
    t3 = timeit(lambda: C.conv_bwd_weight_raw(dy, x, 1, False, 0, dyb=dyb, xb=xb))
    print(f"{Cin:5d}->{Cout:<5d}@{HW:<4d} {gf:8.1f} | {t1:7.1f} {gf / t1 * 1e3:6.1f} | {t2:7.1f} {gf / t2 * 1e3:6.1f} | {t3:7.1f} {gf / t3 * 1e3:6.1f}", flush=True)
