import os, sys, copy, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, uaps_amd
from uaps_amd import conv
DEV = "cuda:0"
rng = np.random.default_rng(21)
def batch(B=2, H=64, W=64):
    xl = torch.tensor(rng.standard_normal((B, 3, H, W)).astype(np.float32)).to(DEV)
    xu = torch.tensor(rng.standard_normal((B, 3, H, W)).astype(np.float32)).to(DEV)
    y = torch.tensor(uaps_amd.data.synthetic_masks(rng, B, 4, H, W)).to(DEV)
    return xl, y, xu
torch.manual_seed(8)
m0 = uaps_amd.UNet_UAPS(3, 4, feature_chns=[8, 16, 16, 32, 32]).to(DEV)
tr = uaps_amd.UAPSTrainer(m0, base_lr=1e-3, seed=5, step_state=True)
def names():
    conv.KERNEL_EVENTS, conv.EVENT_FILTER = {}, None
    tr.train_step(*batch())
    torch.cuda.synchronize()
    c = collections.Counter({k: len(v) for k, v in conv.KERNEL_EVENTS.items()})
    conv.KERNEL_EVENTS = None
    return c
for _ in range(3): tr.train_step(*batch())
a = names()
tr.validate([batch()[:2]])
b = names()
c = names()
print("before validate:", dict(a))
print("after validate :", dict(b))
print("one step later :", dict(c))
