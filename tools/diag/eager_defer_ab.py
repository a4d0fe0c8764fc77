"""Eager single-stream / decoder-stream step with and without the deferred weight-gradient reductions, in one process: host time to
issue a step (train_step returns) and step time with the device.  GPU box: python3 tools/diag/eager_defer_ab.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import uaps_amd
import uaps_amd.unet as _unet
from uaps_amd import conv

dev = torch.device("cuda:0")
for streams in (False, True):
    _unet._DECODER_STREAMS = streams
    torch.manual_seed(1337)
    model = uaps_amd.net_factory("unet_uaps", 3, 4, n_aux=3)
    tr = uaps_amd.UAPSTrainer(model, seed=1337)
    data = uaps_amd.data.SyntheticBatches(16, 3, 4, 256, 256, n_batches=2, seed=1337, device=dev)
    for rep in range(3):
        for defer in (False, True):
            conv._DEFER = defer
            for _ in range(3):
                tr.train_step(*data.next())
            torch.cuda.synchronize()
            host = 0.0
            t0 = time.perf_counter()
            for _ in range(20):
                h0 = time.perf_counter()
                tr.train_step(*data.next())
                host += time.perf_counter() - h0
            torch.cuda.synchronize()
            total = time.perf_counter() - t0
            print(f"streams={streams} defer={defer}: host {host / 20 * 1e3:6.2f} ms/step issued, {total / 20 * 1e3:6.2f} ms/step with the device", flush=True)
