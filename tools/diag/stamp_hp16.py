#!/usr/bin/env python3
"""DIAGNOSTIC (stamps build): cycle table of the persistent 16-output-channel kernels (conv_hp16_body) on the 256 x 256 level, at
B = 32 (tensors partly resident in the Infinity Cache between launches) and B = 128 (nothing resident: what the step sees)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "uaps_amd", "lib", "libuaps_hip_stamps.so")
if os.environ.get("UAPS_HIP_LIB") != LIB:
    os.environ["UAPS_HIP_LIB"] = LIB
    import subprocess
    raise SystemExit(subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], cwd=ROOT).returncode)
import numpy as np
import torch
from uaps_amd import conv as CV, bounds, _lib
PH = ["prologue", "tile 0 fetch+store", "load issue", "matrix loop", "epilogue stores", "barrier", "wait+split+LDS", "barrier"]
dev = torch.device("cuda:0")
L = _lib.lib()
L.uaps_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_ulonglong]
L.uaps_debug_set_stamp_buffer.restype = C.c_int
for B in (32, 128):
    for Cin in (16, 32):
        torch.manual_seed(0)
        x = torch.randn(B, Cin, 256, 256, device=dev)
        w = torch.randn(16, Cin, 3, 3, device=dev) * 0.05
        wf, _ = CV.pack_weights(w)
        xb = (bounds.from_value(x.abs().max()), 1.0)
        for _ in range(20):
            CV.conv_fwd_raw(x, wf, None, 16, 3, 0, xb=xb)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            CV.conv_fwd_raw(x, wf, None, 16, 3, 0, xb=xb)
        e1.record()
        torch.cuda.synchronize()
        per = e0.elapsed_time(e1) / 20 * 1e3
        nw = 1 << 13
        buf = torch.zeros(nw * 20, dtype=torch.int64, device=dev)
        assert L.uaps_debug_set_stamp_buffer(buf.data_ptr(), nw) == 0
        CV.conv_fwd_raw(x, wf, None, 16, 3, 0, xb=xb)
        torch.cuda.synchronize()
        L.uaps_debug_set_stamp_buffer(None, 0)
        d = buf.view(nw, 20).cpu().numpy().astype(np.float64)
        d = d[d[:, 13] > 0]
        cyc = d[:, 13] - d[:, 12]
        wall = (d[:, 15] - d[:, 14]) * 10.0
        clock = np.median(cyc / np.maximum(wall, 10.0))
        span = (d[:, 15].max() - d[:, 14].min()) * 10.0
        ntiles = B * 8 * 32
        nwg = len(d) / 4
        tpw = ntiles / nwg
        ph = d[:, :8].mean(0)
        mb = (B * (Cin + 16) * 65536 * 4) / 1e6
        print(f"{Cin}->16 @256x256 B={B}: {per:.1f} us/launch ({mb / per * 1e-3 * 1e3:.0f} GB/s algorithmic), stamped span {span / 1e3:.1f} us, clock {clock:.2f} GHz, "
              f"{nwg:.0f} workgroups x {tpw:.1f} tiles, wave life {cyc.mean():.0f} cycles = {cyc.mean() / tpw:.0f} per tile")
        print("    per wave: " + ", ".join(f"{n} {v:.0f}" for n, v in zip(PH, ph)))
        print("    per tile: " + ", ".join(f"{n} {v / tpw:.0f}" for n, v in list(zip(PH, ph))[2:]))
        sys.stdout.flush()
