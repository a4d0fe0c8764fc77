import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, uaps_amd
dev = torch.device("cuda:0")
D, B, C, H, W = 4, 16, 4, 256, 256
both = [torch.randn(2 * B, C, H, W, device=dev).mul_(2).requires_grad_(True) for _ in range(D)]
y = torch.randint(0, C, (B, H, W), device=dev)
w = np.random.default_rng(0).dirichlet(np.ones(D))
for _ in range(6):
    uaps_amd.uaps_pair_loss(both, y, w, 0.1, 0.1).loss.backward()
torch.cuda.synchronize()
