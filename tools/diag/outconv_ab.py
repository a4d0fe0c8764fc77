#!/usr/bin/env python3
"""out_conv (16 -> 4 @ 256 x 256, B = 32): the exact-N VALU kernels against the padded matrix-core kernels of round 1 (cfg bit 28)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uaps_amd import conv as C
dev = torch.device("cuda:0")
B, Cin, Cout, HW = 32, 16, 4, 256
x = torch.randn(B, Cin, HW, HW, device=dev); dy = torch.randn(B, Cout, HW, HW, device=dev)
w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
wf, wb = C.pack_weights(w)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for mode in ("exact", "h16"):
    C.set_mode(mode)
    for name, cfg in (("exact-N kernels", 0), ("padded MFMA (round 1 path, cfg bit 28)", 1 << 28)):
        f = t(lambda: C.conv_fwd_raw(x, wf, None, Cout, 3, cfg))
        g = t(lambda: C.conv_bwd_weight_raw(dy, x, 3, True, cfg))
        print(f"mode {mode:5s} {name:42s} fwd {f:7.1f} us   wrw {g:7.1f} us   fwd+wrw {f + g:7.1f} us")
