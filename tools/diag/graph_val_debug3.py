import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, uaps_amd
from uaps_amd import conv, bounds
DEV = "cuda:0"
def batches(n, seed):
    rng = np.random.default_rng(seed); out = []
    for _ in range(n):
        xl = torch.tensor(rng.standard_normal((2, 3, 64, 64)).astype(np.float32)).to(DEV)
        xu = torch.tensor(rng.standard_normal((2, 3, 64, 64)).astype(np.float32)).to(DEV)
        y = torch.tensor(uaps_amd.data.synthetic_masks(rng, 2, 4, 64, 64)).to(DEV)
        out.append((xl, y, xu))
    return out
def run(variant):
    torch.manual_seed(8)
    m0 = uaps_amd.UNet_UAPS(3, 4, feature_chns=[8, 16, 16, 32, 32]); m1 = copy.deepcopy(m0)
    m0.to(DEV); m1.to(DEV)
    A = uaps_amd.UAPSTrainer(m0, base_lr=1e-3, seed=5, step_state=True)
    B = uaps_amd.UAPSTrainer(m1, base_lr=1e-3, seed=5, step_state=True)
    val = [(b[0], b[1]) for b in batches(1, 22)]
    for i, (xl, y, xu) in enumerate(batches(5, 21)):
        for tr in (A, B):
            uaps_amd.perturb.manual_seed(5, 0); np.random.seed(5)
            tr.train_step(xl, y, xu)
        if i == 2:
            if variant == "validate": A.validate(val)
            elif variant == "evaltrain": m0.eval(); m0.train()
            elif variant == "evalfwd":
                m0.eval()
                with torch.no_grad(): m0(val[0][0])
                m0.train()
            elif variant == "trainfwd_nograd":
                with torch.no_grad(): m0(val[0][0])
            elif variant == "invalidate": conv.invalidate_packed_weights()
    torch.cuda.synchronize()
    nd = sum(int(not torch.equal(a, b)) for a, b in zip(m0.parameters(), m1.parameters()))
    print(f"{variant:18s}: differing params after 5 steps: {nd}", flush=True)
for v in ("none", "validate", "evaltrain", "evalfwd", "trainfwd_nograd", "invalidate"):
    run(v)
