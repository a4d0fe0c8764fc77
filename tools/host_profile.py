#!/usr/bin/env python3
"""cProfile of the host side of one training step (where do the ~18 ms of Python/launch time per step go?)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, uaps_amd
dev = torch.device("cuda:0")
model = uaps_amd.net_factory("unet_uaps", 3, 4)
tr = uaps_amd.UAPSTrainer(model, seed=1337)
data = uaps_amd.data.SyntheticBatches(16, 3, 4, 256, 256, n_batches=2, device=dev)
for _ in range(5): tr.train_step(*data.next())
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): tr.train_step(*data.next())
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
