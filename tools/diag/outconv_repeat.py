#!/usr/bin/env python3
"""The launch pair that deviated first in every trace of tools/diag/trace_repeat.py -- uaps_bn_finalize_train + uaps_conv_fwd_bn
on the 8 -> 4 channel out_conv of the small test net (conv_small_bn_kernel<8, 4>) -- repeated on FIXED inputs; every xf and
every z is compared bit for bit with the first.  Run it beside another process that uses the card.
  python tools/diag/outconv_repeat.py REPS [ctx]      ctx: 0 = bare pair, 1 = a split 16 -> 8 convolution in front of every pair"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uaps_amd import conv as CV, _lib, fused

reps = int(sys.argv[1]); ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
torch.manual_seed(3)
B, Cc, H, W, Cout, G = 4, 8, 64, 64, 4, 2
L = _lib.lib()
x = torch.randn(B, Cc, H, W, device=dev)
w1 = torch.randn(Cc, Cc, 3, 3, device=dev) * 0.1
wo = torch.randn(Cout, Cc, 3, 3, device=dev) * 0.1
bo = torch.randn(Cout, device=dev) * 0.1
cb = torch.randn(Cc, device=dev) * 0.1
gamma, beta = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
rm0, rv0 = torch.randn(Cc, device=dev) * 0.1, torch.rand(Cc, device=dev) + 0.5
rm, rv, nbt = rm0.clone(), rv0.clone(), torch.zeros((), dtype=torch.int64, device=dev)
wf1, _ = CV.pack_weights(w1)
wfo, _ = CV.pack_weights(wo)
with fused.stat_groups(G):
    y, st, ppi = CV.conv_fwd_raw(x, wf1, None, Cc, 3, 0, want_stats=True, stat_shift=(rm0, cb))
xc = torch.randn(B, 16, H, W, device=dev); wc = torch.randn(8, 16, 3, 3, device=dev) * 0.1
wfc, _ = CV.pack_weights(wc)
stats = torch.empty((2, G * Cc), dtype=torch.float32, device=dev)


def pair():
    xf = torch.empty((G, Cc, 2), dtype=torch.float32, device=dev)
    z = torch.empty((B, Cout, H, W), dtype=torch.float32, device=dev)
    rm.copy_(rm0); rv.copy_(rv0)
    if ctx:
        CV.conv_fwd_raw(xc, wfc, None, 8, 3, 0, want_stats=True)
    s = _lib.current_stream(dev)
    L.uaps_next_call_hints(_lib.mk_hints((), None, (rm, cb)))
    _lib.check(L.uaps_bn_finalize_train(st.data_ptr(), int(st.shape[2]), cb.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(),
                                        rv.data_ptr(), nbt.data_ptr(), 0.1, 1e-5, B, Cc, H, W, G, stats[0].data_ptr(), stats[1].data_ptr(),
                                        xf.data_ptr(), s), "finalize")
    _lib.check(L.uaps_conv_fwd_bn(y.data_ptr(), xf.data_ptr(), 0.01, G, wfo.data_ptr(), bo.data_ptr(), z.data_ptr(), None, B, Cc, Cout, H, W,
                                  3, 0, s), "conv_fwd_bn")
    return xf, z


xf0, z0 = [t.clone() for t in pair()]
bad_xf = torch.zeros((), dtype=torch.int64, device=dev); bad_z = torch.zeros((), dtype=torch.int64, device=dev)
nbad_elems = torch.zeros((), dtype=torch.int64, device=dev)
maxd = torch.zeros((), device=dev)
t0 = time.time()
events = []
for i in range(reps):
    xf, z = pair()
    if len(events) < 6 and i % 50 == 49:          # (cheap) look for a deviation every 50 launches and keep its geometry
        d = (z.view(torch.int32) != z0.view(torch.int32))
        if bool(d.any()):
            idx = d.nonzero()
            events.append((i, idx.cpu(), (z - z0)[d].cpu(), z[d].cpu(), z0[d].cpu()))
    bad_xf += (xf.view(torch.int32) != xf0.view(torch.int32)).any()
    ne = z.view(torch.int32) != z0.view(torch.int32)
    bad_z += ne.any(); nbad_elems += ne.sum()
    maxd = torch.maximum(maxd, (z - z0).abs().max())
torch.cuda.synchronize()
print(f"pid {os.getpid()} ctx {ctx} mode {CV.get_mode()}: {reps} pairs: xf differed {int(bad_xf)} times, z differed {int(bad_z)} times "
      f"({int(nbad_elems)} elements, max |diff| {float(maxd):.3e}) in {time.time() - t0:.0f}s", flush=True)
for i, idx, dv, zv, z0v in events:
    print(f"launch {i}: {idx.shape[0]} elements differ; (b, c, y, x) -> got / want:")
    for r in range(min(idx.shape[0], 80)):
        print("   ", idx[r].tolist(), f"{float(zv[r]):+.5f} / {float(z0v[r]):+.5f}")
