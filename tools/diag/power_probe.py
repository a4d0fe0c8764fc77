#!/usr/bin/env python3
"""Is the fp16-split forward kernel at the power envelope?  The same launch on random operands and on all-zero operands (same
instructions, same memory traffic, far less switching in the matrix pipe): a power-limited kernel runs faster on zeros."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uaps_amd import conv as C, bounds
dev = torch.device("cuda:0")
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for (Cin, Cout, HW) in ((64, 64, 128), (32, 32, 128), (128, 128, 64)):
    B = 32
    res = []
    for kind in ("random", "zeros", "random"):
        x = torch.randn(B, Cin, HW, HW, device=dev) if kind == "random" else torch.zeros(B, Cin, HW, HW, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05 if kind == "random" else torch.zeros(Cout, Cin, 3, 3, device=dev)
        wf, wb = C.pack_weights(w)
        xb = (bounds.from_value(torch.ones(1, device=dev) * 8.0), 1.0)
        res.append(f"{kind} {t(lambda: C.conv_fwd_raw(x, wf, None, Cout, 3, 0, xb=xb)):.1f} us")
    print(f"{Cin}->{Cout}@{HW} B={B}: " + "   ".join(res), flush=True)
