"""Forward / input-gradient / weight-gradient times of the ResNet-50 bottleneck 1x1 shapes at the configs[4] per-GPU batch (B = 16,
80 x 80 and 160 x 160 maps): python tools/diag/g1_layers.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from uaps_amd import bounds, conv  # noqa: E402

dev = torch.device("cuda:0")
B = 16


def t(fn, it=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


print(f"{'layer':24s} {'GF':>6s} {'MB':>6s} | {'fwd us':>7s} {'TF/s':>6s} {'TB/s':>5s} | {'bwdD':>7s} {'TF/s':>6s} | {'wrw':>7s} {'TF/s':>6s}")
for Cin, Cout, HW in ((256, 64, 160), (64, 256, 160), (512, 128, 80), (128, 512, 80), (1024, 256, 80), (256, 1024, 80), (2048, 512, 80), (512, 2048, 80),
                      (1024, 2048, 80), (2048, 128, 80)):
    x = torch.randn(B, Cin, HW, HW, device=dev)
    w = torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05
    dy = torch.randn(B, Cout, HW, HW, device=dev)
    wf, wb = conv.pack_weights(w)
    xb, dyb = (bounds.from_value(x.abs().max()), 1.0), (bounds.from_value(dy.abs().max()), 1.0)
    gf = 2.0 * B * HW * HW * Cin * Cout / 1e9
    mb = (x.numel() + dy.numel()) * 4 / 1e6
    tf = t(lambda: conv.conv_fwd_raw(x, wf, None, Cout, 1, 0, xb=xb))
    tb = t(lambda: conv.conv_bwd_data_raw(dy, wb, Cin, 1, 0, dyb=dyb))
    tw = t(lambda: conv.conv_bwd_weight_raw(dy, x, 1, False, 0, dyb=dyb, xb=xb))
    print(f"{f'{Cin}->{Cout} @{HW}':24s} {gf:6.1f} {mb:6.0f} | {tf:7.1f} {gf / tf * 1e3:6.1f} {mb / tf:5.2f} | {tb:7.1f} {gf / tb * 1e3:6.1f} | {tw:7.1f} {gf / tw * 1e3:6.1f}")
    del x, w, dy
