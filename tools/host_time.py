#!/usr/bin/env python3
"""How far ahead of the GPU does the host run?  Issue time of K steps (Python + launches, no synchronisation) vs the
time until the GPU has finished them.  If the two are close the step is host-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, uaps_amd
dev = torch.device("cuda:0")
model = uaps_amd.net_factory("unet_uaps", 3, 4)
tr = uaps_amd.UAPSTrainer(model, seed=1337)
data = uaps_amd.data.SyntheticBatches(16, 3, 4, 256, 256, n_batches=2, device=dev)
for _ in range(5): tr.train_step(*data.next())
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K): tr.train_step(*data.next())
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {1e3 * (t1 - t0) / K:.2f} ms/step, until GPU done {1e3 * (t2 - t0) / K:.2f} ms/step")
