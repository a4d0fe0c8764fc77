import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, uaps_amd
dev = torch.device("cuda:0")
model = uaps_amd.net_factory("unet_uaps", 3, 4)
tr = uaps_amd.UAPSTrainer(model, seed=1337)
data = uaps_amd.data.SyntheticBatches(2, 3, 4, 32, 32, n_batches=2, device=dev)
for _ in range(5): tr.train_step(*data.next())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): tr.train_step(*data.next())
torch.cuda.synchronize()
print(f"host-bound step (2+2 images 32x32): {1e3 * (time.perf_counter() - t0) / 30:.2f} ms")
