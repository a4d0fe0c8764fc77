#!/usr/bin/env python3
"""DIAGNOSTIC (needs the stamp build of conv_s32_body): per-phase cycles of the h32 kernel, averaged over waves."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uaps_amd import conv as CV, bounds, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
L.uaps_debug_set_conv_buffer.argtypes = [C.c_void_p]
for (Cin, Cout, HW) in ((64, 64, 128), (32, 32, 128), (128, 128, 64)):
    B = 32
    x = torch.randn(B, Cin, HW, HW, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    wf, wb = CV.pack_weights(w)
    xb = (bounds.from_value(x.abs().max()), 1.0)
    nblk = 1 << 16
    dbg = torch.zeros(nblk * 4 * 8, dtype=torch.int64, device=dev)
    for it in range(3):
        if it == 2:
            L.uaps_debug_set_conv_buffer(dbg.data_ptr())
        CV.conv_fwd_raw(x, wf, None, Cout, 3, 0, xb=xb)
        torch.cuda.synchronize()
    L.uaps_debug_set_conv_buffer(None)
    d = dbg.view(nblk, 4, 8).cpu().double()
    used = d[:, :, 6] > 0
    n = int(used.sum())
    m = d[used].mean(0)
    names = ["prologue", "load issue", "mfma loop", "barrier1", "store", "barrier2", "total"]
    t0 = d[used][:, 7]
    span = float((t0 + d[used][:, 6]).max() - t0.min())
    print(f"{Cin}->{Cout}@{HW}: waves {n}  kernel span {span:.0f} cyc  per-wave: " + "  ".join(f"{names[i]} {m[i]:.0f}" for i in range(7)))
