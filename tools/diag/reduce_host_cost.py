"""Host cost of the weight-gradient calls (per call, launches queued without synchronising): partial, reduce, and the pair, on a
decoder-sized layer.  GPU box: python3 tools/diag/reduce_host_cost.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from uaps_amd import _lib, bounds, conv

dev = torch.device("cuda:0")
L = _lib.lib()
B, Cin, Cout, H, W, ks = 32, 64, 64, 64, 64, 3
x = torch.randn(B, Cin, H, W, device=dev)
dy = torch.randn(B, Cout, H, W, device=dev)
xb, dyb = (bounds.from_value(x.abs().max()), 1.0), (bounds.from_value(dy.abs().max()), 1.0)
cfg = conv.plan_cfg(ks, 0, True, dy, x)
n = C.c_size_t()
L.uaps_conv_wrw_workspace_bytes(B, Cin, Cout, H, W, ks, cfg, C.byref(n))
ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
dw = torch.empty(Cout, Cin, ks, ks, device=dev)
st = _lib.current_stream(dev)


def timed(fn, reps=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6


def partial():
    L.uaps_next_call_hints(_lib.mk_hints((dyb, xb)))
    L.uaps_conv_bwd_weight_partial(dy.data_ptr(), x.data_ptr(), 0, B, Cin, Cout, H, W, ks, cfg, ws.data_ptr(), ws.numel(), st)


def reduce():
    L.uaps_conv_bwd_weight_reduce(ws.data_ptr(), dw.data_ptr(), None, B, Cin, Cout, H, W, ks, cfg, st)


for name, fn in (("partial", partial), ("reduce", reduce), ("partial + reduce", lambda: (partial(), reduce())),
                 ("conv_bwd_weight_raw", lambda: conv.conv_bwd_weight_raw(dy, x, ks, False, 0, dyb=dyb, xb=xb))):
    for _ in range(3):
        fn()
    host, total = timed(fn)
    print(f"{name:24s} host {host:7.1f} us per call, with the device {total:7.1f} us")
