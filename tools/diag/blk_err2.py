import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import torch.nn.functional as F
from test_res_uaps import _backbone
from uaps_amd import bounds, conv, res_uaps
dev = torch.device("cuda:0")
conv.set_mode("exact")
cpu = _backbone().train()
gpu = _backbone().to(dev).train()
bc, bg = cpu.layer2[0], gpu.layer2[0]
conv.pack_all([m.weight for m in gpu.modules() if isinstance(m, torch.nn.Conv2d) and m.kernel_size[0] in (1, 3)])
g = torch.Generator().manual_seed(256)
x = torch.relu(torch.randn(2, 256, 64, 64, generator=g))
xg = x.to(dev)
e = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max())
with torch.no_grad():
    for name, cv_c, cv_g, inp in (("conv1", bc.conv1, bg.conv1, x), ("conv2", bc.conv2, bg.conv2, torch.relu(torch.randn(2, 128, 64, 64, generator=g))),
                                  ("conv3", bc.conv3, bg.conv3, torch.relu(torch.randn(2, 128, 32, 32, generator=g))), ("down", bc.downsample[0], bg.downsample[0], x)):
        yc = cv_c(inp)
        ig = inp.to(dev)
        if cv_g.stride == (1, 1):
            yg = conv.conv2d(ig, cv_g.weight, None)
            yg2, st = conv.conv2d_with_stats(ig, cv_g.weight, None)
            print(name, "conv2d", e(yg, yc), "with_stats", e(yg2, yc), "stats sum", float(st[..., 0].sum()), float(yc.sum()))
        elif cv_g.kernel_size == (1, 1):
            sub = conv.subsample2(ig)
            yg = conv.conv2d(sub, cv_g.weight, None)
            yg2, st = conv.conv2d_with_stats(sub, cv_g.weight, None)
            print(name, "sub+conv2d", e(yg, yc), "with_stats", e(yg2, yc), "stats sum", float(st[..., 0].sum()), float(yc.sum()))
        else:
            yg = conv.conv2d_strided(ig, cv_g.weight, 2, 1)
            print(name, "strided", e(yg, yc))
    print("block y", e(bg(xg), bc(x)))
    for name, cv_c, bn_c, cv_g, bn_g, relu, inp in (("c1", bc.conv1, bc.bn1, bg.conv1, bg.bn1, True, x),):
        print(name, e(res_uaps.conv_bn_act(xg, cv_g, bn_g, relu, True), F.relu(bn_c(cv_c(x)))))
