#!/usr/bin/env python3
"""Runs one convolution direction of one layer a few times (for rocprofv3 --pmc): layer_run.py Cin Cout HW ks cfg mode [dir] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uaps_amd import conv as C
Cin, Cout, HW, ks, cfg = (int(v, 0) for v in sys.argv[1:6])
mode = sys.argv[6]
d = sys.argv[7] if len(sys.argv) > 7 else "fwd"
B = int(sys.argv[8]) if len(sys.argv) > 8 else 32
dev = torch.device("cuda:0")
C.set_mode(mode)
x = torch.randn(B, Cin, HW, HW, device=dev)
w = torch.randn(Cout, Cin, ks, ks, device=dev) * 0.05
dy = torch.randn(B, Cout, HW, HW, device=dev)
wf, wb = C.pack_weights(w)
from uaps_amd import bounds
xb = (bounds.from_value(x.abs().max()), 1.0) if mode == "h16" else None
dyb = (bounds.from_value(dy.abs().max()), 1.0) if mode == "h16" else None
for _ in range(6):
    if d == "fwd":
        C.conv_fwd_raw(x, wf, None, Cout, ks, cfg, xb=xb)
    elif d == "bwd":
        C.conv_bwd_data_raw(dy, wb, Cin, ks, cfg, dyb=dyb)
    else:
        C.conv_bwd_weight_raw(dy, x, ks, False, cfg, dyb=dyb, xb=xb)
torch.cuda.synchronize()
