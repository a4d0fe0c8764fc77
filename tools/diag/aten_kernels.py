"""Which library (ATen) kernels still run in a training step, with counts and device time: python tools/diag/aten_kernels.py
[--net resnet50_uaps] [--size 96] [--batch 2] [--classes 2] [--stacks].  The product path is meant to be hand-written kernels only."""
import argparse
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import uaps_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--net", default="resnet50_uaps")
ap.add_argument("--size", type=int, default=96)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--classes", type=int, default=2)
ap.add_argument("--stacks", action="store_true")
a = ap.parse_args()
torch.manual_seed(0)
model = uaps_amd.net_factory(a.net, 3, a.classes, n_aux=3)
tr = uaps_amd.UAPSTrainer(model, base_lr=1e-4)
data = uaps_amd.data.SyntheticBatches(a.batch, 3, a.classes, a.size, a.size, n_batches=1, device="cuda:0")
xl, yl, xu = data.next()
for _ in range(2):
    tr.train_step(xl, yl, xu)
torch.cuda.synchronize()
acts = [ProfilerActivity.CUDA] + ([ProfilerActivity.CPU] if a.stacks else [])
with profile(activities=acts, with_stack=a.stacks) as prof:
    tr.train_step(xl, yl, xu)
    torch.cuda.synchronize()
tot = 0.0
rows = []
for e in prof.key_averages(group_by_stack_n=8 if a.stacks else 0):
    dt = getattr(e, "device_time_total", 0.0) or getattr(e, "cuda_time_total", 0.0)
    if dt <= 0:
        continue
    tot += dt
    rows.append((dt, e.count, e.key, getattr(e, "stack", None)))
rows.sort(reverse=True)
print(f"device time of one step: {tot / 1e3:.3f} ms")
lib = [r for r in rows if "at::native" in r[2] or "at::" in r[2] or "Memcpy" in r[2] or "Memset" in r[2]]
print(f"library kernels: {sum(r[0] for r in lib) / 1e3:.3f} ms in {sum(r[1] for r in lib)} launches")
for dt, n, k, st in lib:
    print(f"{dt / 1e3:9.3f} ms {n:5d}  {k[:150]}")
    if st:
        for s in st[:8]:
            print("            ", s)
