"""Per-block forward/backward error of the ResNet bottlenecks on the HIP path vs CPU module math, in every conv mode."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from test_res_uaps import _backbone
from uaps_amd import bounds, conv
dev = torch.device("cuda:0")
for layer, idx, cin, hw in [("layer2", 0, 256, 64), ("layer3", 0, 512, 32), ("layer3", 1, 1024, 32), ("layer4", 1, 2048, 32)]:
    cpu = _backbone().train()
    blk_c = getattr(cpu, layer)[idx]
    g = torch.Generator().manual_seed(cin + idx)
    x = torch.relu(torch.randn(2, cin, hw, hw, generator=g))
    xc = x.clone().requires_grad_(True)
    yc = blk_c(xc)
    dy = torch.randn(yc.shape, generator=g)
    yc.backward(dy)
    xd = x.double().requires_grad_(True)
    blk_d = getattr(_backbone().train().double(), layer)[idx]
    yd = blk_d(xd); yd.backward(dy.double())
    pe = max(float((pc.grad - pd.grad).norm() / pd.grad.norm()) for pc, pd in zip(blk_c.parameters(), blk_d.parameters()))
    print("cpu params L2", pe)
    print(layer, idx, "cpu fp32 vs fp64: y %.2e dx %.2e" % (float((yc - yd).norm() / yd.norm()), float((xc.grad - xd.grad).norm() / xd.grad.norm())))
    for mode in ("h16", "split", "exact"):
        conv.set_mode(mode)
        gpu = _backbone().to(dev).train()
        blk_g = getattr(gpu, layer)[idx]
        bns = [m for m in gpu.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        conv.pack_all([m.weight for m in gpu.modules() if isinstance(m, torch.nn.Conv2d) and m.kernel_size[0] in (1, 3)])
        bounds.refresh(bns)
        xg = x.to(dev).requires_grad_(True)
        yg = blk_g(bounds.put(xg, bounds.from_value(xg.detach().abs().max()), 1.0))
        dyg = dy.to(dev)
        yg.backward(bounds.put(dyg, bounds.from_value(dyg.abs().max()), 1.0))
        e = lambda a, b: float((a.cpu().double() - b).norm() / b.norm())
        pe = max(e(pg.grad, pd.grad) for pg, pd in zip(blk_g.parameters(), blk_d.parameters()))
        print("   ", mode, "vs fp64: y %.2e dx %.2e params %.2e" % (e(yg.detach(), yd.detach()), e(xg.grad, xd.grad), pe))
