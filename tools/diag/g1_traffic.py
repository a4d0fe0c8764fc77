"""One GEMM-tiled 1x1 layer per run, for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (tools/diag/g1_traffic.sh): does conv_g1h_kernel re-read
its input?  python tools/diag/g1_traffic.py Cin Cout HW [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from uaps_amd import bounds, conv  # noqa: E402

Cin, Cout, HW = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(B, Cin, HW, HW, device=dev)
w = torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05
wf, wb = conv.pack_weights(w)
xb = (bounds.from_value(x.abs().max()), 1.0)
for _ in range(3):
    y = conv.conv_fwd_raw(x, wf, None, Cout, 1, 0, xb=xb)
torch.cuda.synchronize()
print("algorithmic MB", (x.numel() + y.numel()) * 4 / 1e6, "in", x.numel() * 4 / 1e6, "out", y.numel() * 4 / 1e6)
