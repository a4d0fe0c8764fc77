#!/usr/bin/env python3
"""Experiment: speed and error of the s32 forward kernel in its current mode (UAPS_SPLIT_F16=0: three bf16 pieces, =1: two fp16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from uaps_amd import conv as C
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (Cin, Cout, HW, B) in ((32, 32, 256, 32), (64, 64, 128, 32), (128, 128, 64, 32), (256, 256, 32, 32), (32, 64, 128, 32)):
    x = torch.randn(B, Cin, HW, HW, device=dev)
    x = torch.where(x > 0, x, 0.01 * x)                       # LeakyReLU-shaped activations
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (9 * Cin)) ** 0.5
    wf, wb = C.pack_weights(w)
    for _ in range(3):
        y = C.conv_fwd_raw(x, wf, None, Cout, 3, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = C.conv_fwd_raw(x, wf, None, Cout, 3, 0)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ye = C.conv_fwd_raw(x, wf, None, Cout, 3, 1 << 28)        # exact fp32 MFMA
    ref = F.conv2d(x[:1].double().cpu(), w.double().cpu(), padding=1)
    err = (y[:1].double().cpu() - ref).abs().max().item()
    erre = (ye[:1].double().cpu() - ref).abs().max().item()
    rms = ((y[:1].double().cpu() - ref) ** 2).mean().sqrt().item()
    rmse = ((ye[:1].double().cpu() - ref) ** 2).mean().sqrt().item()
    fl = 2.0 * B * HW * HW * Cin * Cout * 9
    print(f"{Cin:4d}->{Cout:4d} @{HW:3d}  {us:8.1f} us  {fl / us / 1e6:7.1f} TF   max err {err:.2e} (exact kernel {erre:.2e})  rms {rms:.2e} (exact {rmse:.2e})  ref scale {ref.abs().max():.2f}", flush=True)
