#!/usr/bin/env python3
"""DIAGNOSTIC (stamps build: `make -C uaps_amd/csrc stamps`): the cycle table of the fp16-split forward kernel (conv_s32_body).

  python tools/diag/stamp_table.py            runs itself once per library variant (shipped body + the ablations e5 / e6 / e7)

Per layer and variant: launches the layer back to back for ~2 s on random data (the clock the chip holds under that load), then
stamps ONE launch: every wave records the shader cycles (s_memtime) it spent per phase, its first and last stamp and the 100 MHz
wall clock (s_memrealtime) at both ends.  Printed: kernel span (wall), in-kernel clock = cycles / wall, launch skew and tail,
mean cycles per phase per wave, cycles per MFMA inside the matrix loop (back-to-back constant: 32 for 32x32x16 with one wave per
SIMD, 64 when two waves share the pipe and keep it full), and the share of the span in which a SIMD's matrix pipe has work.
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIBDIR = os.path.join(ROOT, "uaps_amd", "lib")
VARIANTS = [("shipped body", "libuaps_hip_stamps.so"), ("e5: no staging behind chunk 0", "libuaps_hip_stamps_e5.so"),
            ("e6: + no barriers", "libuaps_hip_stamps_e6.so"), ("e7: MFMAs only", "libuaps_hip_stamps_e7.so")]
LAYERS = [(64, 64, 64), (32, 32, 128), (128, 128, 32)]      # (Cin, Cout, H = W), B = 32: conv_h32_kernel<64>, conv_h32t_kernel<32>, conv_h32_kernel<64>
PHASES = ["prologue", "chunk 0 fetch+store", "load issue", "matrix loop", "barrier", "LDS stores", "barrier", "epilogue stores", "statistics"]


def child():
    import numpy as np
    import torch
    from uaps_amd import conv as CV, bounds, _lib
    dev = torch.device("cuda:0")
    L = _lib.lib()
    L.uaps_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_ulonglong]
    L.uaps_debug_set_stamp_buffer.restype = C.c_int
    secs = float(os.environ.get("UAPS_STAMP_WARM_S", "2.0"))
    for Cin, Cout, HW in LAYERS:
        B = 32
        torch.manual_seed(0)
        x = torch.randn(B, Cin, HW, HW, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
        wf, _ = CV.pack_weights(w)
        xb = (bounds.from_value(x.abs().max()), 1.0)
        name = CV.kernel_variant("fwd", B, Cin, Cout, HW, HW, 3).replace("conv_s32", "conv_h32")
        y = CV.conv_fwd_raw(x, wf, None, Cout, 3, 0, xb=xb)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            CV.conv_fwd_raw(x, wf, None, Cout, 3, 0, xb=xb)
        e1.record()
        torch.cuda.synchronize()
        per = e0.elapsed_time(e1) / 200 * 1e3
        n = int(secs * 1e6 / per)
        for _ in range(n):                                   # back-to-back load: the clock the part settles at
            CV.conv_fwd_raw(x, wf, None, Cout, 3, 0, xb=xb)
        nw = 1 << 16
        buf = torch.zeros(nw * 20, dtype=torch.int64, device=dev)
        assert L.uaps_debug_set_stamp_buffer(buf.data_ptr(), nw) == 0
        CV.conv_fwd_raw(x, wf, None, Cout, 3, 0, xb=xb)
        torch.cuda.synchronize()
        assert L.uaps_debug_set_stamp_buffer(None, 0) == 0
        d = buf.view(nw, 20).cpu().numpy().astype(np.float64)
        d = d[d[:, 13] > 0]
        cyc = d[:, 13] - d[:, 12]                            # a wave's life in shader cycles
        wall = (d[:, 15] - d[:, 14]) * 10.0                  # ns
        clock = np.median(cyc / np.maximum(wall, 10.0))      # GHz
        span_ns = (d[:, 15].max() - d[:, 14].min()) * 10.0
        skew_ns = (d[:, 14].max() - d[:, 14].min()) * 10.0
        tail_ns = (d[:, 15].max() - np.median(d[:, 15])) * 10.0
        ph = d[:, :9].mean(0)
        nchunks = (Cin + 7) // 8
        bn = 64 if "<64>" in name else 32
        mr = 4 if "h32t" in name else 2
        mfma = 5 * mr * 3 * (bn // 32)                       # per wave and chunk
        first = np.median(d[:, 17] - d[:, 12])
        print(f"{Cin}->{Cout} @{HW}x{HW} B={B} {name}: {len(d)} waves, {per:.1f} us/launch back to back; stamped launch: span {span_ns / 1e3:.1f} us, "
              f"in-kernel clock {clock:.2f} GHz, start skew {skew_ns / 1e3:.2f} us, tail (last end - median end) {tail_ns / 1e3:.2f} us")
        print("    cycles per wave: " + ", ".join(f"{n} {v:.0f}" for n, v in zip(PHASES, ph)) + f"; life {cyc.mean():.0f}; first matrix loop starts {first:.0f} after the wave")
        loop = ph[3] / (nchunks * mfma)
        busy = len(d) / 1024.0 * nchunks * mfma * 32 / (span_ns * clock)   # all waves' MFMAs (32 cycles each) over the chip's 1024 SIMDs
        print(f"    matrix loop: {mfma} MFMAs per wave and chunk x {nchunks} chunks -> {loop:.1f} cycles per MFMA of this wave (32 = pipe to itself; x waves per SIMD "
              f"= {len(d) / 1024.0:.1f} when they share a full pipe); matrix work of a SIMD / span = {busy:.2f}")
        sys.stdout.flush()


def main():
    if os.environ.get("UAPS_STAMP_CHILD"):
        return child()
    for label, lib in VARIANTS:
        path = os.path.join(LIBDIR, lib)
        if not os.path.exists(path):
            print(f"== {label}: {lib} not built (make -C uaps_amd/csrc stamps)")
            continue
        print(f"== {label} ({lib})", flush=True)
        env = dict(os.environ, UAPS_HIP_LIB=path, UAPS_STAMP_CHILD="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        print(r.stdout, end="")
        if r.returncode != 0:
            print(r.stderr[-2000:])


if __name__ == "__main__":
    main()
