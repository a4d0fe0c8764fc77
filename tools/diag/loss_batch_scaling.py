"""Is the loss forward's 3.8-4.2 TB/s a streaming problem or the fixed cost of a 35 us launch?  (round 6)
The one-launch loss block at D = C = 4, 256 x 256 for B = 16 (the step's shape; the launches rotate over four input sets = 604 MB, so
nothing is served from the 256 MiB Infinity Cache), 32, 64 and 128 images per branch: dispatch-event time of pair_fwd / pair_bwd and
the rate on their algorithmic bytes.  If the rate rises with B, the kernel streams fine and the launch's fixed part (ramp, first
loads, reduction epilogue) is what the step's shape pays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import uaps_amd
from uaps_amd import losses

dev = torch.device("cuda:0")
D, C, H, W = 4, 4, 256, 256
w = np.random.default_rng(0).dirichlet(np.ones(D))
for B, nsets in ((16, 4), (32, 2), (64, 1), (128, 1)):
    sets = []
    for s in range(nsets):
        both = [torch.randn(2 * B, C, H, W, device=dev).mul_(2).requires_grad_(True) for _ in range(D)]
        y = torch.randint(0, C, (B, H, W), device=dev)
        sets.append((both, y))
    N = B * H * W
    fb, bb = N * 2 * (4 * D * C + 8), N * 2 * (8 * D * C + 8)
    tf, tb = [], []
    for it in range(24):
        both, y = sets[it % nsets]
        losses.KERNEL_EVENTS = {}
        out = uaps_amd.uaps_pair_loss(both, y, w, 0.1, 0.1)
        out.loss.backward()
        torch.cuda.synchronize()
        ev = losses.KERNEL_EVENTS
        if it >= 4:
            tf.append(ev["uaps_pair_fwd"][0][0].elapsed_time(ev["uaps_pair_fwd"][0][1]) * 1e3)
            tb.append(ev["uaps_pair_bwd"][0][0].elapsed_time(ev["uaps_pair_bwd"][0][1]) * 1e3)
    losses.KERNEL_EVENTS = None
    f, b = float(np.median(tf)), float(np.median(tb))
    print(f"B = {B:3d} per branch ({nsets} input set(s), {fb / 1e6:6.0f} MB read per forward): pair_fwd {f:7.1f} us = {fb / f / 1e6:5.2f} TB/s | "
          f"pair_bwd {b:7.1f} us = {bb / b / 1e6:5.2f} TB/s", flush=True)
    del sets
    torch.cuda.empty_cache()
