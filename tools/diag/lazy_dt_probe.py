import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, torch.nn as nn
from uaps_amd import _lib, bounds, conv, fused, lazybn
from uaps_amd.fused import _bn_ws
DEV = torch.device("cuda:0")
for (B, Cin, C, H, W) in ((8, 32, 32, 128, 128), (4, 32, 32, 32, 64), (8, 64, 64, 64, 64)):
    torch.manual_seed(1)
    groups = 2
    x = torch.randn(B, Cin, H, W, device=DEV)
    y = torch.randn(B, C, H, W, device=DEV) * 2 + 0.3
    dout = torch.randn(B, C, H, W, device=DEV)
    bn = nn.BatchNorm2d(C).to(DEV)
    stats = torch.empty((2, groups * C), device=DEV)
    out = torch.empty_like(y)
    ws = _bn_ws(DEV, B, C, H, W)
    L = _lib.lib()
    with _lib.device_guard(DEV):
        rc = L.uaps_bn_act_fwd_train_grouped(y.data_ptr(), None, bn.weight.data_ptr(), bn.bias.data_ptr(), None, None, None, 0.1, bn.eps, 0.01, 0.0, 0, 0,
                                             B, C, H, W, groups, out.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), ws.data_ptr(), ws.numel(), _lib.current_stream(DEV))
    dg, db, dc = (torch.empty(C, device=DEV) for _ in range(3))
    xb = (bounds.from_value(x.abs().max()), 1.0)
    res = []
    for rep in range(3):
        lz = lazybn.prepare(dout, y, bn.weight, bn.bias, stats[0], stats[1], 0.01, groups, dg, db, dc, ws)
        lazybn.take(dout)
        ref = lazybn.materialize(dout, lz)
        dw, _, dyt = conv.conv_bwd_weight_raw(dout, x, 3, False, 0, xb=xb, lz=lz)
        torch.cuda.synchronize()
        same = torch.equal(dyt, ref)
        bad = (dyt != ref)
        print((B, Cin, C, H, W), "rep", rep, "dy equal", same, "mismatches", int(bad.sum()), "max err", float((dyt - ref).abs().max()),
              "where", bad.nonzero()[:3].tolist() if not same else "")
        res.append(dw.clone())
    print("   dw deterministic:", all(torch.equal(r, res[0]) for r in res))
