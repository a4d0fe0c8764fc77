"""ResNet-encoder variant (uaps_amd/res_uaps.py, BASELINE.json configs[4]): the backbone is
pinned against the reference's utilities/resnet.py through fixture g7_resnet.npz (formula weights, no RNG); the UAPS
decoder on top of it is this build's design (the reference has none) and is covered by behaviour tests."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from formula_weights import formula_state_dict


def _backbone():
    import uaps_amd
    net = uaps_amd.res_uaps.resnet50()
    net.load_state_dict(formula_state_dict(net.state_dict()))
    return net


def test_resnet50_state_dict_layout_equals_reference():
    g = np.load(os.path.join(GOLDEN, "g7_resnet.npz"))
    sd = _backbone().state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g["shapes"]]


def test_resnet50_cpu_module_math_vs_reference_fixture():
    g = np.load(os.path.join(GOLDEN, "g7_resnet.npz"))
    net, x = _backbone().eval(), torch.tensor(g["x"])
    with torch.no_grad():
        for i, c in enumerate(net.base_forward(x)):
            np.testing.assert_allclose(c.numpy(), g[f"eval_c{i + 1}"], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_resnet50_hip_path_vs_reference_fixture():
    g = np.load(os.path.join(GOLDEN, "g7_resnet.npz"))
    dev = torch.device("cuda:0")
    net, x = _backbone().to(dev), torch.tensor(g["x"], device=dev)
    net.eval()
    with torch.no_grad():
        for i, c in enumerate(net.base_forward(x)):
            ref = g[f"eval_c{i + 1}"]
            np.testing.assert_allclose(c.cpu().numpy(), ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max())
    # Train mode: with 2 x 6 x 6 samples per channel the batch statistics amplify rounding differences by ~3x per block
    # (measured: 4e-5 after layer1.0, O(1) after layer3), so the end-to-end comparison stops after layer1 ...
    net.train()
    with torch.no_grad():
        cs = net.base_forward(x)
    ref = g["train_c1"]
    np.testing.assert_allclose(cs[0].cpu().numpy(), ref, rtol=5e-3, atol=5e-4 * np.abs(ref).max())
    # ... and the deeper (dilated) blocks are compared one at a time on identical inputs against the plain-PyTorch
    # module math of the same class on the CPU (itself pinned to the reference by the eval-mode fixture above).
    cpu = _backbone().train()
    feat = torch.tensor(g["eval_c3"])
    for blk_g, blk_c in ((net.layer4[0], cpu.layer4[0]), (net.layer4[1], cpu.layer4[1])):
        with torch.no_grad():
            o_c = blk_c(feat)
            o_g = blk_g(feat.to(dev)).cpu()
        assert float((o_c - o_g).abs().max()) <= 2e-4 * float(o_c.abs().max())
        feat = o_c


@pytest.mark.gpu
@pytest.mark.parametrize("layer,idx,cin,hw", [("layer2", 0, 256, 64), ("layer3", 0, 512, 32), ("layer3", 1, 1024, 32), ("layer4", 0, 1024, 32),
                                              ("layer4", 1, 2048, 32)])
@pytest.mark.parametrize("mode", ["h16", "split", "exact"])
def test_bottleneck_block_forward_backward_vs_cpu_module_math(layer, idx, cin, hw, mode):
    """One Bottleneck (utilities/resnet.py:55-95) in train mode, forward and every gradient, against the plain-PyTorch module
    math of the same class on the CPU: layer2.0 holds the two strided convolutions and the sampled shortcut projection,
    layer3 / layer4 the dilated (2 / 4) 3x3 convolutions.  Inputs are ReLU outputs with a magnitude bound, as inside the net."""
    from uaps_amd import bounds, conv
    dev = torch.device("cuda:0")
    cpu = _backbone().train()
    gpu = _backbone().to(dev).train()
    blk_c, blk_g = getattr(cpu, layer)[idx], getattr(gpu, layer)[idx]
    g = torch.Generator().manual_seed(cin + idx)
    x = torch.relu(torch.randn(2, cin, hw, hw, generator=g))
    xc = x.clone().requires_grad_(True)
    yc = blk_c(xc)
    dy = torch.randn(yc.shape, generator=g)
    yc.backward(dy)
    prev = conv.get_mode()
    conv.set_mode(mode)
    try:
        gpu._bns = [m for m in gpu.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        conv.pack_all([m.weight for m in gpu.modules() if isinstance(m, torch.nn.Conv2d) and m.kernel_size[0] in (1, 3)])
        bounds.refresh(gpu._bns)
        xg = x.to(dev).requires_grad_(True)
        yg = blk_g(bounds.put(xg, bounds.from_value(xg.detach().abs().max()), 1.0))
        dyg = dy.to(dev)
        yg.backward(bounds.put(dyg, bounds.from_value(dyg.abs().max()), 1.0))
    finally:
        conv.set_mode(prev)

    # Forward: rounding only.  Gradients: BatchNorm outputs that sit within rounding of zero flip their ReLU mask between any
    # two fp32 evaluations (PyTorch-CPU fp32 against float64 differs by 1e-4 ... 4e-3 in relative L2 norm on these blocks,
    # tools/diag/blk_err.py); each operator's own gradient is pinned tightly in test_gpu_conv / test_gpu_strided / test_gpu_fused,
    # this test pins the wiring (a wrong operand, layout or missing term is an O(1) error).
    def close(a, b, what, tol):
        err = float((a.cpu() - b).norm() / (b.norm() + 1e-20))
        assert err <= tol, f"{layer}[{idx}] {what}: relative L2 error {err:.3e}"

    close(yg.detach(), yc.detach(), "y", 1e-4)
    close(xg.grad, xc.grad, "dx", 4e-2)
    for (n, pc), (_, pg) in zip(blk_c.named_parameters(), blk_g.named_parameters()):
        close(pg.grad, pc.grad, n, 4e-2)


@pytest.mark.gpu
def test_res_uaps_training_steps():
    """Whole UAPS steps on the ResNet-50 encoder variant through the product path (forward_pair, grouped BatchNorm,
    pair loss, HIP Adam): finite, decreasing loss and a finite gradient for every parameter."""
    import uaps_amd
    torch.manual_seed(0)
    model = uaps_amd.net_factory("resnet50_uaps", 3, 2, n_aux=3)
    assert isinstance(model, uaps_amd.ResUAPS)
    tr = uaps_amd.UAPSTrainer(model, base_lr=1e-4)
    assert tr.pair_forward
    data = uaps_amd.data.SyntheticBatches(2, 3, 2, 96, 96, n_batches=1, device="cuda:0")
    xl, yl, xu = data.next()
    seen = {}
    hooks = [p.register_hook(lambda gr, n=n: seen.__setitem__(n, bool(torch.isfinite(gr).all()))) for n, p in model.named_parameters()]
    losses_seen = [float(tr.train_step(xl, yl, xu)["loss"]) for _ in range(8)]
    for h in hooks:
        h.remove()
    assert all(np.isfinite(losses_seen)), losses_seen
    assert min(losses_seen[-3:]) < losses_seen[0], losses_seen
    assert len(seen) == len(list(model.parameters())) and all(seen.values())
    out = model.eval()(xl)
    assert len(out) == 4 and out[0].shape == (2, 2, 96, 96)
