#!/usr/bin/env python3
"""Which host call sites launch ATen fill / copy kernels inside a training step (they should be near zero)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, uaps_amd
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
model = uaps_amd.net_factory("unet_uaps", 3, 4)
tr = uaps_amd.UAPSTrainer(model, seed=1337)
data = uaps_amd.data.SyntheticBatches(2, 3, 4, 64, 64, n_batches=2, device=dev)
for _ in range(3): tr.train_step(*data.next())
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    tr.train_step(*data.next())
torch.cuda.synchronize()
from collections import Counter
c = Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::add", "aten::add_", "aten::ones_like", "aten::zeros_like", "aten::mul"):
        st = [s for s in (ev.stack or []) if "uaps_amd" in s or "torch/autograd" in s][:2]
        c[(ev.name, tuple(st), str(ev.input_shapes)[:60])] += 1
for k, v in c.most_common(30):
    print(v, k)
