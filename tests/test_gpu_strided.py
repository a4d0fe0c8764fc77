"""-m gpu: the general strided convolution kernels and the 3x3 / 2 max-pool (csrc/conv_strided.hip) that take the last
library ops off the ResNet-encoder path (utilities/resnet.py:120, 124, 147, 8-14), against PyTorch on the CPU; then one
whole UAPS step of the ResNet-50 variant at BASELINE.json configs[4]'s per-GPU shape (8 + 8 images of 640 x 640)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (B, Cin, Cout, H, W, ks, stride, pad)
CASES = [
    (2, 3, 64, 96, 96, 7, 2, 3),          # the stem
    (2, 3, 64, 50, 70, 7, 2, 3),          # odd sizes
    (1, 1, 32, 64, 64, 7, 2, 3),          # grey-scale stem: one input channel (the stem weight-gradient kernel, 49 of 160 columns)
    (2, 3, 80, 40, 72, 7, 2, 3),          # ... two blocks of output channels, partial tiles
    (2, 3, 64, 320, 320, 7, 2, 3),
    (2, 128, 128, 40, 40, 3, 2, 1),       # layer2.0.conv2
    (2, 256, 512, 40, 40, 1, 2, 0),       # layer2.0.downsample
    (1, 20, 24, 17, 23, 3, 2, 1),
    (2, 16, 40, 19, 21, 5, 1, 2),         # forward / input gradient only (5x5 has no weight-gradient kernel)
    (2, 8, 8, 16, 16, 3, 1, 1),
    (1, 64, 64, 320, 320, 1, 2, 0),
]


def _mk(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32)


@pytest.mark.parametrize("B,Cin,Cout,H,W,ks,stride,pad", CASES)
def test_strided_conv_vs_torch_cpu(B, Cin, Cout, H, W, ks, stride, pad):
    from uaps_amd.conv import conv2d_strided
    x = _mk((B, Cin, H, W), 1)
    w = _mk((Cout, Cin, ks, ks), 2) / np.sqrt(Cin * ks * ks)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, pad)
    dy = _mk(tuple(yr.shape), 3)
    yr.backward(dy.double())
    want_dw = ks in (1, 3, 7)
    xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(want_dw)
    y = conv2d_strided(xg, wg, stride, pad)
    assert y.shape == yr.shape
    y.backward(dy.to(DEV))

    def close(a, ref, what):
        scale = float(ref.abs().max()) + 1e-12
        err = float((a.detach().cpu().double() - ref).abs().max())
        assert err <= 2e-5 * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"

    close(y, yr.detach(), "y")
    close(xg.grad, xr.grad, "dx")
    if want_dw:
        close(wg.grad, wr.grad, "dw")


@pytest.mark.parametrize("B,C,H,W", [(2, 64, 48, 48), (1, 3, 7, 9), (2, 5, 33, 20), (4, 64, 320, 320)])
def test_maxpool3x3s2_vs_torch_cpu(B, C, H, W):
    from uaps_amd.conv import maxpool3x3s2
    x = torch.relu(_mk((B, C, H, W), 5))            # ReLU output: many exact ties (zeros), like the stem
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    dy = _mk(tuple(yr.shape), 6)
    yr.backward(dy)
    xg = x.to(DEV).requires_grad_(True)
    y = maxpool3x3s2(xg)
    y.backward(dy.to(DEV))
    assert torch.equal(y.detach().cpu(), yr.detach())
    assert torch.equal(xg.grad.cpu(), xr.grad)       # ties resolved like torch: the first maximum of the window


@pytest.mark.parametrize("B,C,H,W", [(2, 16, 48, 64), (1, 3, 7, 9), (2, 5, 33, 20), (2, 256, 160, 160), (1, 4, 8, 12)])
def test_subsample2_and_its_adjoint(B, C, H, W):
    """x[:, :, ::2, ::2] (the sampling of the 1x1 / stride-2 shortcut, utilities/resnet.py:157-161) and its adjoint: bit-exact
    copies; both the 16-byte and the scalar form."""
    from uaps_amd.conv import subsample2
    x = _mk((B, C, H, W), 15).to(DEV).requires_grad_(True)
    y = subsample2(x)
    ref = x.detach()[:, :, ::2, ::2]
    assert y.shape == ref.shape and torch.equal(y.detach(), ref)
    dy = _mk(tuple(y.shape), 16).to(DEV)
    y.backward(dy)
    want = torch.zeros_like(x.detach())
    want[:, :, ::2, ::2] = dy
    assert torch.equal(x.grad, want)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 128, 128, 40, 40), (1, 16, 24, 18, 24), (2, 64, 32, 64, 64), (1, 8, 16, 6, 8)])
@pytest.mark.parametrize("mode", ["h16", "split", "exact"])
def test_conv3x3_stride2_over_sampling_phases(B, Cin, Cout, H, W, mode):
    """layer2.0.conv2 (utilities/resnet.py:8-10 with stride 2): the stride-1 convolution over the four sampling phases
    (uaps_space_to_depth2 + re-arranged weights) against F.conv2d(stride=2, padding=1) in float64: forward, input and weight
    gradient; and the phase tensor itself, bit for bit, with its inverse."""
    from uaps_amd import _lib, bounds, conv
    x = torch.relu(_mk((B, Cin, H, W), 41))
    w = _mk((Cout, Cin, 3, 3), 42) / np.sqrt(Cin * 9)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 2, 1)
    dy = _mk(tuple(yr.shape), 43)
    yr.backward(dy.double())
    xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    xs = conv._SpaceToDepth2.apply(xg.detach())
    want = torch.stack([xg.detach()[:, :, py::2, px::2] for py in (0, 1) for px in (0, 1)], dim=1).reshape(B, 4 * Cin, H // 2, W // 2)
    assert torch.equal(xs, want)
    back = torch.empty_like(xg.detach())
    _lib.check(_lib.lib().uaps_space_to_depth2(xs.data_ptr(), back.data_ptr(), B, Cin, H, W, 1, _lib.current_stream(xs.device)), "inverse")
    assert torch.equal(back, xg.detach())
    prev = conv.get_mode()
    conv.set_mode(mode)
    try:
        y, st = conv.conv3x3s2(bounds.put(xg, bounds.from_value(xg.detach().abs().max()), 1.0), wg, with_stats=True)
        dyg = dy.to(DEV)
        y.backward(bounds.put(dyg, bounds.from_value(dyg.abs().max()), 1.0))
    finally:
        conv.set_mode(prev)

    def close(a, ref, what):
        scale = float(ref.abs().max()) + 1e-12
        err = float((a.detach().cpu().double() - ref).abs().max())
        assert err <= 2e-5 * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"

    close(y, yr.detach(), "y")
    close(xg.grad, xr.grad, "dx")
    close(wg.grad, wr.grad, "dw")
    np.testing.assert_allclose(st[..., 0].double().sum((1, 2)).cpu().numpy(), yr.detach().sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)


def test_no_library_convolution_or_pooling_on_the_resnet_path():
    """The ResNet-50 encoder on the GPU must not call F.conv2d / max-pool: run its forward + backward under a profiler-free
    check -- torch's convolution and pooling entry points are patched to raise."""
    import uaps_amd
    net = uaps_amd.res_uaps.resnet50().to(DEV).train()
    x = torch.randn(2, 3, 96, 96, device=DEV)

    def boom(*a, **k):
        raise AssertionError("library convolution / pooling called on the GPU path")

    saved = (F.conv2d, F.max_pool2d, torch.conv2d, torch.max_pool2d)
    F.conv2d = F.max_pool2d = boom
    torch.conv2d = torch.max_pool2d = boom
    try:
        cs = net.base_forward(x)
        sum(c.sum() for c in cs).backward()
    finally:
        F.conv2d, F.max_pool2d, torch.conv2d, torch.max_pool2d = saved
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


def test_no_library_elementwise_add_in_the_resnet_backward():
    """Residual joins hand one handle per consumer to the next block (res_uaps.add_relu): the gradients of a join's consumers are
    summed inside its own backward pass (uaps_relu_bwd_sum), not by the autograd engine's accumulation -- which is an ATen
    elementwise-add launch per join (20 per step of the ResNet-50 net, 3.6 ms at the configs[4] shape in round 3).  A training
    step of the ResNet-50 UAPS net under the profiler must show no such kernel."""
    import uaps_amd
    try:
        from torch.profiler import ProfilerActivity, profile
    except Exception as e:                                   # pragma: no cover
        pytest.skip(f"torch.profiler unavailable: {e}")
    torch.manual_seed(0)
    model = uaps_amd.net_factory("resnet50_uaps", 3, 2, n_aux=3)
    tr = uaps_amd.UAPSTrainer(model, base_lr=1e-4)
    data = uaps_amd.data.SyntheticBatches(2, 3, 2, 96, 96, n_batches=1, device=DEV)
    xl, yl, xu = data.next()
    tr.train_step(xl, yl, xu)
    torch.cuda.synchronize()
    try:
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            tr.train_step(xl, yl, xu)
            torch.cuda.synchronize()
        counts = {e.key: e.count for e in prof.key_averages()}
    except Exception as e:                                   # pragma: no cover
        pytest.skip(f"device profiling not available here: {e}")
    if not any("relu_bwd_sum_kernel" in n for n in counts):
        pytest.skip("the profiler recorded no device kernels of this package")
    # one scalar add belongs to the trainer's bookkeeping; the net has 16 joins, each of which was at least one add launch
    adds = sum(c for n, c in counts.items() if "CUDAFunctor_add" in n)
    assert adds <= 2, {n: c for n, c in counts.items() if "CUDAFunctor_add" in n}
    joins = sum(c for n, c in counts.items() if "relu_bwd_sum_kernel" in n)
    assert joins == 16, joins                                # 3 + 4 + 6 + 3 Bottleneck blocks


def test_res_uaps_step_at_config4_shape():
    """BASELINE.json configs[4] per-GPU shape: ResNet-50 encoder, K = 3, 2 classes, 640 x 640, 8 labelled + 8 unlabelled
    images: steps through the product path; finite, decreasing loss, a finite gradient for every parameter."""
    import uaps_amd
    torch.manual_seed(0)
    model = uaps_amd.net_factory("resnet50_uaps", 3, 2, n_aux=3)
    tr = uaps_amd.UAPSTrainer(model, base_lr=1e-4)
    data = uaps_amd.data.SyntheticBatches(8, 3, 2, 640, 640, n_batches=1, device=DEV)
    xl, yl, xu = data.next()
    seen = {}
    hooks = [p.register_hook(lambda gr, n=n: seen.__setitem__(n, bool(torch.isfinite(gr).all()))) for n, p in model.named_parameters()]
    losses_seen = [float(tr.train_step(xl, yl, xu)["loss"]) for _ in range(4)]
    for h in hooks:
        h.remove()
    assert all(np.isfinite(losses_seen)), losses_seen
    assert min(losses_seen[1:]) < losses_seen[0], losses_seen
    assert len(seen) == len(list(model.parameters())) and all(seen.values())
