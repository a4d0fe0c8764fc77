import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F, copy
from uaps_amd import conv, fused
DEV = "cuda:0"
B, Cin, Cout, H, W, ks = 16, 16, 2, 512, 512, 3
g = torch.Generator().manual_seed(B + Cin + Cout + H + ks)
x0 = torch.randn(B, 8, H, W, generator=g)
w0 = torch.randn(Cin, 8, 3, 3, generator=g) / np.sqrt(72.0)
bn = torch.nn.BatchNorm2d(Cin)
with torch.no_grad():
    bn.weight.uniform_(0.5, 1.5, generator=g); bn.bias.uniform_(-0.5, 0.5, generator=g)
w2 = torch.randn(Cout, Cin, ks, ks, generator=g) / np.sqrt(Cin * ks * ks)
b2 = torch.randn(Cout, generator=g)
dz = torch.randn(B, Cout, H, W, generator=g)
# 1. plain bwd data
wf, wb = conv.pack_weights(w2.to(DEV))
da = conv.conv_bwd_data_raw(dz.to(DEV), wb, Cin, ks)
da_ref = torch.nn.grad.conv2d_input((B, Cin, H, W), w2, dz, padding=1)
e = (da.cpu() - da_ref).abs()
print("bwd_data max err", float(e.max()), "scale", float(da_ref.abs().max()))
if float(e.max()) > 1e-3:
    idx = torch.nonzero(e > 1e-3)
    print("n bad", idx.shape[0], "first", idx[:5].tolist(), "last", idx[-5:].tolist())
    print("bad per image", torch.bincount(idx[:, 0], minlength=B).tolist())
    print("bad per channel", torch.bincount(idx[:, 1], minlength=Cin).tolist())
# 2. bn backward alone, with da_ref
bng = copy.deepcopy(bn).to(DEV)
with fused.stat_groups(2):
    y, st = conv.conv2d_with_stats(x0.to(DEV), w0.to(DEV), None)
    y = y.detach().requires_grad_(True)
    a = fused.bn_act(y, None, bng, 0.01, 0.0, True, st)
a.backward(da_ref.to(DEV))
yc = y.detach().cpu().requires_grad_(True)
h = B // 2
ac = torch.cat([F.leaky_relu(bn(yc[:h]), 0.01), F.leaky_relu(bn(yc[h:]), 0.01)], 0)
ac.backward(da_ref)
e = (y.grad.cpu() - yc.grad).abs()
print("bn bwd max err", float(e.max()), "scale", float(yc.grad.abs().max()))
print("act err", float((a.detach().cpu() - ac.detach()).abs().max()))
if float(e.max()) > 1e-3:
    idx = torch.nonzero(e > 1e-3)
    print("n bad", idx.shape[0], "first", idx[:5].tolist())
    print("bad per image", torch.bincount(idx[:, 0], minlength=B).tolist())
    print("bad per channel", torch.bincount(idx[:, 1], minlength=Cin).tolist())
