"""The U-Net's four 1x1 projections (UAPS_unet.py:73) at the step's launch shape (B = 32): the plan's kernel (fp32 MFMA, 8-channel
chunks: 19.7 us for the 13.7 MB of 256 -> 128 @16^2 in profiles/r05_final_kernel_stats.csv -- bound by neither HBM nor the matrix
pipe but by 32 serial chunk rounds in 256 workgroups) against the forced split form (cfg bit 29: 16x16x32 MFMA on fp16 pieces,
32-channel chunks = a quarter of the rounds).  Forward, input gradient, weight gradient; back-to-back launches, one stream.  (round 6)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uaps_amd import conv as C, bounds

C.set_mode("h16")
dev = torch.device("cuda:0")
B = 32
bnd = lambda t: (bounds.from_value(t.abs().max()), 1.0)


def timeit(fn, iters=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print("layer (B = 32)        MB moved | plan: fwd  dgrad   wrw us | forced split form: fwd  dgrad   wrw us")
for Cin, Cout, HW in ((256, 128, 16), (128, 64, 32), (64, 32, 64), (32, 16, 128)):
    x = torch.randn(B, Cin, HW, HW, device=dev)
    w = torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05
    dy = torch.randn(B, Cout, HW, HW, device=dev)
    wf, wb = C.pack_weights(w)
    xb, dyb = bnd(x), bnd(dy)
    mb = 4.0 * B * HW * HW * (Cin + Cout) / 1e6
    row = []
    for cfg in (0, 1 << 29):
        try:
            t1 = timeit(lambda: C.conv_fwd_raw(x, wf, None, Cout, 1, cfg, xb=xb))
            t2 = timeit(lambda: C.conv_bwd_data_raw(dy, wb, Cin, 1, cfg, dyb=dyb))
            t3 = timeit(lambda: C.conv_bwd_weight_raw(dy, x, 1, False, cfg, dyb=dyb, xb=xb))
            row.append(f"{t1:7.1f} {t2:7.1f} {t3:7.1f}")
        except Exception as e:
            row.append(f"({type(e).__name__}: {str(e)[:60]})")
    # parity of the forced form against the plan's (both fp32-accurate)
    y0 = C.conv_fwd_raw(x, wf, None, Cout, 1, 0, xb=xb)
    y1 = C.conv_fwd_raw(x, wf, None, Cout, 1, 1 << 29, xb=xb)
    err = float((y0 - y1).abs().max() / y0.abs().max())
    print(f"{Cin:4d} -> {Cout:<4d}@{HW:<4d} {mb:8.1f} | {row[0]} | {row[1]}   max rel diff {err:.1e}", flush=True)
