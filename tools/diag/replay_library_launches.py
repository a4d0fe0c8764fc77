#!/usr/bin/env python3
"""Which library launches does a REPLAYED training step still make, and from where?  Profiles one replay (CPU + GPU activities, Python
stacks) with the batch already in the graph's own input tensors and the live output scalars, and prints every aten:: operator that
launched device work with its innermost Python frames.  GPU box: python tools/diag/replay_library_launches.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

import uaps_amd

dev = "cuda:0"
torch.manual_seed(9)
model = uaps_amd.UNet_UAPS(3, 4, feature_chns=[8, 16, 16, 32, 32]).to(dev)
tr = uaps_amd.UAPSTrainer(model, base_lr=1e-3, seed=0, use_graph=True, track_metrics=False)
rng = np.random.default_rng(5)
xl = torch.tensor(rng.standard_normal((4, 3, 64, 64)).astype(np.float32)).to(dev)
xu = torch.tensor(rng.standard_normal((4, 3, 64, 64)).astype(np.float32)).to(dev)
y = torch.tensor(uaps_amd.data.synthetic_masks(rng, 4, 4, 64, 64)).to(dev)
for _ in range(4):
    tr.train_step(xl, y, xu)
g = tr.step_graph
g.live_outputs = "--clones" not in sys.argv
xs = g.inputs() if "--own-inputs" not in sys.argv else (xl, y, xu)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(*xs)
    torch.cuda.synchronize()
n_own = 0
for e in prof.events():
    name = str(e.name)
    if e.device_type.name == "CUDA":
        if "uaps::" in name or "(anonymous namespace)::" in name:
            n_own += 1
        else:
            print(f"device: {name[:90]:90s} {getattr(e, 'device_time', 0):8.1f} us")
    elif name.startswith("aten::") and (getattr(e, "device_time", 0) or 0) > 0:
        st = [q for q in (e.stack or []) if "uaps_amd" in q or "bench" in q][:4]
        print(f"host:   {name[:40]:40s} device time {e.device_time:6.1f} us   {' <- '.join(st)}")
print(f"{n_own} launches of the library's own kernels")
