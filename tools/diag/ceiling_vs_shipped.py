"""The shipped fp16-split forward kernels on the four layers of tools/split_gemm_ceiling.hip, same protocol (B = 32, random
LeakyReLU-shaped data, bounds supplied, >= 0.5 s of back-to-back launches before 50 timed ones, one stream), so that probe and
kernel are read from ONE box.  Usage: python tools/diag/ceiling_vs_shipped.py [ceiling.json]   (round 6)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uaps_amd import conv as C, bounds

C.set_mode("h16")
dev = torch.device("cuda:0")
B = 32
ceil = {}
if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
    ceil = {l["layer"]: l for l in json.load(open(sys.argv[1]))["layers"]}
print("layer (B = 32)        GFLOP | shipped kernel                         us   TFLOP/s | probe us (best, A re-read)  TFLOP/s | shipped / probe")
for name, Cin, Cout, HW in (("64 -> 64 @64^2", 64, 64, 64), ("128 -> 64 @64^2", 128, 64, 64), ("128 -> 128 @32^2", 128, 128, 32),
                            ("256 -> 128 @32^2", 256, 128, 32)):
    x = torch.randn(B, Cin, HW, HW, device=dev)
    x = torch.where(x > 0, x * 3.0, x * 0.03)
    w = (torch.rand(Cout, Cin, 3, 3, device=dev) * 2 - 1) * 0.05
    wf, _ = C.pack_weights(w)
    xb = (bounds.from_value(x.abs().max()), 1.0)
    fn = lambda: C.conv_fwd_raw(x, wf, None, Cout, 3, 0, xb=xb)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    while True:
        for _ in range(100):
            fn()
        e.record(); e.synchronize()
        if s.elapsed_time(e) > 500:
            break
    s.record()
    for _ in range(50):
        fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 50 * 1e3
    gf = 2.0 * B * HW * HW * Cin * Cout * 9 / 1e9
    kern = C.kernel_variant("fwd", B, Cin, Cout, HW, HW, 3, 0).replace("conv_s32", "conv_h32").replace("conv_sfwd", "conv_hfwd")
    c = ceil.get(name)
    tail = f"{gf / c['best_tflops_reread'] * 1e3:10.1f} {c['best_tflops_reread']:18.1f} | {gf / us * 1e3 / c['best_tflops_reread']:.2f}" if c else ""
    print(f"{name:18s} {gf:8.2f} | {kern:36s} {us:7.1f} {gf / us * 1e3:8.1f} | {tail}", flush=True)
