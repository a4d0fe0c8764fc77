#!/usr/bin/env python3
"""Micro-benchmark: plain conv kernels vs the variants that apply BatchNorm + LeakyReLU while staging
(uaps_conv_fwd_bn / uaps_conv_bwd_weight_partial_bn) at the decoder conv2 shapes of the bench step (B = 32).
Run on the GPU box:  python tools/bench_bnconv.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uaps_amd import _lib, conv

SHAPES = [(128, 128, 32, 3), (64, 64, 64, 3), (32, 32, 128, 3), (16, 16, 256, 3), (128, 64, 32, 1), (16, 4, 256, 3)]


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    B, G = 32, 2
    L = _lib.lib()
    st = _lib.current_stream(dev)
    for Cin, Cout, HW, ks in SHAPES:
        x = torch.randn(B, Cin, HW, HW, device=dev)
        dy = torch.randn(B, Cout, HW, HW, device=dev)
        w = torch.randn(Cout, Cin, ks, ks, device=dev) * 0.05
        wf, wb = conv.pack_weights(w)
        y = torch.empty(B, Cout, HW, HW, device=dev)
        xf = torch.rand(G, Cin, 2, device=dev)
        n = C.c_size_t()
        L.uaps_conv_wrw_workspace_bytes(B, Cin, Cout, HW, HW, ks, 0, C.byref(n))
        ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
        args = (B, Cin, Cout, HW, HW, ks, 0)
        t0 = timeit(lambda: L.uaps_conv_fwd(x.data_ptr(), wf.data_ptr(), None, y.data_ptr(), *args, st))
        t1 = timeit(lambda: L.uaps_conv_fwd_bn(x.data_ptr(), xf.data_ptr(), 0.01, G, wf.data_ptr(), None, y.data_ptr(), None, *args, st))
        t2 = timeit(lambda: L.uaps_conv_bwd_weight_partial(dy.data_ptr(), x.data_ptr(), 0, *args, ws.data_ptr(), ws.numel(), st))
        t3 = timeit(lambda: L.uaps_conv_bwd_weight_partial_bn(dy.data_ptr(), x.data_ptr(), xf.data_ptr(), 0.01, G, 0, *args, ws.data_ptr(),
                                                              ws.numel(), st))
        print(f"{Cin:4d}->{Cout:4d}@{HW:3d} k{ks} | fwd {t0:7.1f} fwd_bn {t1:7.1f} ({(t1 / t0 - 1) * 100:+5.1f}%) | wrw {t2:7.1f} wrw_bn {t3:7.1f} "
              f"({(t3 / t2 - 1) * 100:+5.1f}%)", flush=True)


if __name__ == "__main__":
    main()
