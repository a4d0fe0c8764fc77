#!/usr/bin/env python3
"""Does a matrix-core convolution kernel run faster after an idle gap?  The same launch (32 -> 32 channels, 3x3, 32 images
256 x 256, fp16-split kernel conv_h32_kernel<32>) timed on its dispatch (uaps_next_launch_events) back to back, and with the
GPU left idle for a while before every launch.   python tools/diag/gap_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from uaps_amd import _lib, bounds
from uaps_amd.conv import conv2d


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(32, 32, 256, 256, device=dev)
    x = bounds.put(x, bounds.from_value(x.abs().max()), 1.0)
    w = torch.randn(32, 32, 3, 3, device=dev) * 0.05
    for _ in range(20):
        conv2d(x, w)
    torch.cuda.synchronize()
    for gap_ms in (0.0, 0.05, 0.2, 1.0, 5.0):
        ts = []
        for rep in range(3):
            timers = []
            for i in range(40):
                if gap_ms:
                    torch.cuda.synchronize()
                    time.sleep(gap_ms * 1e-3)
                with _lib.LaunchTimer() as t:
                    conv2d(x, w)
                timers.append(t)
            torch.cuda.synchronize()
            ts += [t.elapsed_ms() * 1e3 for t in timers[5:]]
        print(f"idle gap {gap_ms:5.2f} ms before each launch: kernel {np.median(ts):7.2f} us (p10 {np.percentile(ts, 10):7.2f}, p90 {np.percentile(ts, 90):7.2f})", flush=True)


if __name__ == "__main__":
    main()
