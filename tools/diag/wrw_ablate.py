#!/usr/bin/env python3
"""The fp16-split weight-gradient tile kernel ALONE (partial launch, no reduction) on the step's four dominant layers at B = 32, back to
back after half a second of warm-up -- the same layers, data shape and timing recipe as tools/split_wrw_ceiling.hip, so that the
shipped kernel, its ablations (UAPS_HIP_LIB=uaps_amd/lib/libuaps_hip_wrwe<N>.so, make -C uaps_amd/csrc wrwabl) and the geometry-free
probe can be read side by side.  GPU box:  python tools/diag/wrw_ablate.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from uaps_amd import _lib, bounds, conv

LAYERS = [("64 -> 64 @64^2", 64, 64, 64), ("128 -> 64 @64^2", 128, 64, 64), ("128 -> 128 @32^2", 128, 128, 32), ("256 -> 128 @32^2", 256, 128, 32)]


def main():
    dev = torch.device("cuda:0")
    B = 32
    conv.set_mode("h16")
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(7)
    print(f"library: {_lib.LIB_PATH}")
    for name, Cin, Cout, HW in LAYERS:
        x = torch.rand(B, Cin, HW, HW, generator=g) * 2 - 1
        x = torch.where(x > 0, x * 3.0, x * 0.03).to(dev)
        dy = ((torch.rand(B, Cout, HW, HW, generator=g) * 2 - 1) * 0.01).to(dev)
        xb, dyb = (bounds.from_value(x.abs().max()), 1.0), (bounds.from_value(dy.abs().max()), 1.0)
        n = C.c_size_t()
        _lib.check(L.uaps_conv_wrw_workspace_bytes(B, Cin, Cout, HW, HW, 3, 0, C.byref(n)), "ws")
        ws = torch.empty(n.value // 4, dtype=torch.float32, device=dev)
        st = _lib.current_stream(dev)

        def launch():
            rc = L.uaps_conv_bwd_weight_partial_h(_lib.mk_hints((dyb, xb)), dy.data_ptr(), x.data_ptr(), 0, B, Cin, Cout, HW, HW, 3, 0,
                                                  ws.data_ptr(), n.value, st)
            _lib.check(rc, "partial")

        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            for _ in range(100):
                launch()
            e1.record(); e1.synchronize()
            if e0.elapsed_time(e1) > 500.0:
                break
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            launch()
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        gf = 2.0 * B * HW * HW * Cin * Cout * 9 / 1e9
        print(f"  {name:18s} {gf:6.2f} GFLOP  {us:7.1f} us  {gf / us * 1e3:6.1f} TFLOP/s   {conv.kernel_variant('wrw', B, Cin, Cout, HW, HW, 3, 0)}", flush=True)


if __name__ == "__main__":
    main()
