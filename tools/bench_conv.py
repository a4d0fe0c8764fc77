#!/usr/bin/env python3
"""Micro-benchmark of the hand-written convolution kernels against MIOpen (torch.nn.functional.conv2d)
at the layer shapes of UNet_UAPS(3, 4) with 16 images of 256x256 (BASELINE.json configs[1]).
Run on the GPU box:  python tools/bench_conv.py [--cfg-sweep]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from uaps_amd import conv as C

# (name, Cin, Cout, HW, ks, calls per forward of the 4-head net)
LAYERS = [
    ("enc.in0 3->16@256", 3, 16, 256, 3, 1), ("enc.in1 16->16@256", 16, 16, 256, 3, 1),
    ("enc.d1a 16->32@128", 16, 32, 128, 3, 1), ("enc.d1b 32->32@128", 32, 32, 128, 3, 1),
    ("enc.d2a 32->64@64", 32, 64, 64, 3, 1), ("enc.d2b 64->64@64", 64, 64, 64, 3, 1),
    ("enc.d3a 64->128@32", 64, 128, 32, 3, 1), ("enc.d3b 128->128@32", 128, 128, 32, 3, 1),
    ("enc.d4a 128->256@16", 128, 256, 16, 3, 1), ("enc.d4b 256->256@16", 256, 256, 16, 3, 1),
    ("dec.up1.1x1 256->128@16", 256, 128, 16, 1, 4), ("dec.up1a 256->128@32", 256, 128, 32, 3, 4),
    ("dec.up1b 128->128@32", 128, 128, 32, 3, 4),
    ("dec.up2.1x1 128->64@32", 128, 64, 32, 1, 4), ("dec.up2a 128->64@64", 128, 64, 64, 3, 4),
    ("dec.up2b 64->64@64", 64, 64, 64, 3, 4),
    ("dec.up3.1x1 64->32@64", 64, 32, 64, 1, 4), ("dec.up3a 64->32@128", 64, 32, 128, 3, 4),
    ("dec.up3b 32->32@128", 32, 32, 128, 3, 4),
    ("dec.up4.1x1 32->16@128", 32, 16, 128, 1, 4), ("dec.up4a 32->16@256", 32, 16, 256, 3, 4),
    ("dec.up4b 16->16@256", 16, 16, 256, 3, 4), ("dec.out 16->4@256", 16, 4, 256, 3, 4),
]


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--cfg-sweep", action="store_true")
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--mode", default="h16", choices=["h16", "split", "exact"],
                    help="convolution arithmetic (uaps_amd.conv.set_mode); h16 passes the operands' magnitude bounds like the step does")
    ap.add_argument("--no-miopen", action="store_true", help="skip the MIOpen / ATen columns (large batches)")
    ap.add_argument("--lds-sweep", action="store_true", help="forward kernel with 0..56 KB of extra (unused) LDS = fewer workgroups per CU")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.batch
    C.set_mode(args.mode)
    from uaps_amd import bounds
    bnd = (lambda t: (bounds.from_value(t.abs().max()), 1.0)) if args.mode == "h16" else (lambda t: None)
    tot = {k: 0.0 for k in ("fwd", "bwd", "wrw", "mfwd", "mbwd", "mwrw")}
    print(f"{'layer':28s} {'GF':>6s} | {'fwd us':>8s} {'TF/s':>6s} {'miopen':>8s} | {'bwdD us':>8s} {'TF/s':>6s} {'miopen':>8s} | {'wrw us':>8s} {'TF/s':>6s} {'miopen':>8s}")
    for name, Cin, Cout, HW, ks, calls in LAYERS:
        if args.only and not any(o in name for o in args.only.split(",")):
            continue
        x = torch.randn(B, Cin, HW, HW, device=dev)
        w = torch.randn(Cout, Cin, ks, ks, device=dev) * 0.05
        dy = torch.randn(B, Cout, HW, HW, device=dev)
        wf, wb = C.pack_weights(w)
        gf = 2.0 * B * HW * HW * Cin * Cout * ks * ks / 1e9
        xb, dyb = bnd(x), bnd(dy)
        t_f = timeit(lambda: C.conv_fwd_raw(x, wf, None, Cout, ks, xb=xb))
        t_b = timeit(lambda: C.conv_bwd_data_raw(dy, wb, Cin, ks, dyb=dyb))
        t_w = timeit(lambda: C.conv_bwd_weight_raw(dy, x, ks, False, dyb=dyb, xb=xb))
        if args.no_miopen:
            m_f = m_b = m_w = 0.0
        else:
            m_f = timeit(lambda: F.conv2d(x, w, None, padding=ks // 2))
            m_b = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [ks // 2] * 2, [1, 1], False, [0, 0], 1, [True, False, False]))
            m_w = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [ks // 2] * 2, [1, 1], False, [0, 0], 1, [False, True, False]))
        for k, v in zip(("fwd", "bwd", "wrw", "mfwd", "mbwd", "mwrw"), (t_f, t_b, t_w, m_f, m_b, m_w)):
            tot[k] += v * calls
        print(f"{name:28s} {gf:6.2f} | {t_f:8.1f} {gf / t_f * 1e3:6.1f} {m_f:8.1f} | {t_b:8.1f} {gf / t_b * 1e3:6.1f} {m_b:8.1f} | {t_w:8.1f} {gf / t_w * 1e3:6.1f} {m_w:8.1f}", flush=True)
        if args.lds_sweep:
            res = []
            for rep in range(2):
                for kb in (0, 8, 16, 24, 32, 40, 56):
                    t = timeit(lambda: C.conv_fwd_raw(x, wf, None, Cout, ks, cfg=kb << 16), iters=20)
                    res.append(f"{kb}KB:{t:.1f}")
            print("      fwd extra-LDS sweep (us): " + " ".join(res))
        if args.cfg_sweep:
            for bn in (16, 32, 64):
                for tile in (1, 2):
                    try:
                        t = timeit(lambda: C.conv_fwd_raw(x, wf, None, Cout, ks, cfg=bn | (tile << 8)), iters=20)
                        print(f"      fwd bn={bn} tile={'8x32' if tile == 1 else '16x16'}: {t:8.1f} us {gf / t * 1e3:6.1f} TF/s")
                    except Exception as ex:
                        pass
            for ns in (64, 128, 256, 512, 1024, 2048):
                try:
                    t = timeit(lambda: C.conv_bwd_weight_raw(dy, x, ks, False, cfg=ns))
                    print(f"      wrw nsplit={ns}: {t:8.1f} us {gf / t * 1e3:6.1f} TF/s")
                except Exception as ex:
                    pass
    print("per forward+backward of one 16-image batch through the 4-head net (us): "
          + ", ".join(f"{k}={v:.0f}" for k, v in tot.items()))


if __name__ == "__main__":
    main()
