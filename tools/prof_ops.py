import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, uaps_amd
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
model = uaps_amd.net_factory("unet_uaps", 3, 4)
tr = uaps_amd.UAPSTrainer(model)
data = uaps_amd.data.SyntheticBatches(4, 3, 4, 64, 64, n_batches=1, device=dev)
for _ in range(2): tr.train_step(*data.next())
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as p:
    tr.train_step(*data.next())
torch.cuda.synchronize()
from collections import Counter
c = Counter()
for e in p.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::add", "aten::add_", "aten::copy_", "aten::clone", "aten::mul", "aten::sum"):
        par = e.cpu_parent.name if e.cpu_parent is not None else "-"
        gp = e.cpu_parent.cpu_parent.name if (e.cpu_parent is not None and e.cpu_parent.cpu_parent is not None) else "-"
        c[(e.name, par, gp, str(e.input_shapes)[:60])] += 1
for k, v in c.most_common(40): print(v, k)
