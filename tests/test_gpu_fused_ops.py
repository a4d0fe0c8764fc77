"""-m gpu: the ConvBlock/UpBlock glue kernels (BN+LeakyReLU+Dropout, bilinear-x2+concat) against a
plain PyTorch fp32 reference of the same ops (the reference's own modules: nn.BatchNorm2d,
nn.LeakyReLU, nn.Dropout, nn.Upsample(align_corners=True), torch.cat -- UAPS_unet.py:36-44, 72-86)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref_bn_act(y, bias, bn, slope, train):
    z = F.batch_norm(y + bias.view(1, -1, 1, 1), bn.running_mean, bn.running_var, bn.weight, bn.bias, training=train,
                     momentum=bn.momentum, eps=bn.eps)
    return F.leaky_relu(z, slope)


@pytest.mark.parametrize("shape", [(4, 16, 64, 64), (3, 5, 7, 9), (2, 64, 16, 16), (16, 16, 256, 256), (2, 256, 2, 2)])
def test_bn_act_train_vs_torch(shape):
    from uaps_amd import fused
    torch.manual_seed(1)
    B, C, H, W = shape
    y = (torch.randn(shape, device=DEV) * 1.7 + 0.3).requires_grad_(True)
    bias = torch.randn(C, device=DEV).requires_grad_(True)
    bn = nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2)
    bn_ref = nn.BatchNorm2d(C).to(DEV); bn_ref.load_state_dict(bn.state_dict())
    out = fused.bn_act(y, bias, bn, 0.01, 0.0, True)
    yr = y.detach().clone().requires_grad_(True); br = bias.detach().clone().requires_grad_(True)
    bn_ref.num_batches_tracked += 1
    ref = _ref_bn_act(yr, br, bn_ref, 0.01, True)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(bn.running_mean, bn_ref.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var, bn_ref.running_var, rtol=1e-4, atol=1e-6)
    assert int(bn.num_batches_tracked) == 1
    g = torch.randn_like(out)
    out.backward(g); ref.backward(g)
    scale = float(yr.grad.abs().max())
    # at the LeakyReLU kink (|z| ~ 1 ulp) the slope choice may legitimately differ: exclude those pixels
    z = F.batch_norm(yr.detach() + br.detach().view(1, -1, 1, 1), None, None, bn_ref.weight, bn_ref.bias, training=True, eps=bn.eps)
    ok = z.abs() > 1e-5
    assert float((~ok).float().mean()) < 1e-4
    chan_ok = ok.all(dim=3).all(dim=2).all(dim=0)          # channels without a kink pixel
    ok = ok & chan_ok.view(1, -1, 1, 1)
    torch.testing.assert_close(y.grad[ok], yr.grad[ok], rtol=1e-3, atol=2e-5 * scale)
    torch.testing.assert_close(bn.weight.grad[chan_ok], bn_ref.weight.grad[chan_ok], rtol=1e-4, atol=1e-4 * float(bn_ref.weight.grad.abs().max()))
    torch.testing.assert_close(bn.bias.grad[chan_ok], bn_ref.bias.grad[chan_ok], rtol=1e-4, atol=1e-4 * float(bn_ref.bias.grad.abs().max()))
    assert torch.count_nonzero(bias.grad) == 0                     # exact zero; torch's is rounding noise
    assert float(br.grad.abs().max()) < 1e-3 * float(g.abs().sum() / C)


def test_bn_act_eval_vs_torch():
    from uaps_amd import fused
    torch.manual_seed(2)
    y = torch.randn(3, 8, 20, 12, device=DEV, requires_grad=True)
    bias = torch.randn(8, device=DEV)
    bn = nn.BatchNorm2d(8).to(DEV).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2)
    rm0 = bn.running_mean.clone()
    out = fused.bn_act(y, bias, bn, 0.01, 0.3, False)             # dropout is off in eval
    yr = y.detach().clone().requires_grad_(True)
    ref = _ref_bn_act(yr, bias, bn, 0.01, False)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    assert torch.equal(bn.running_mean, rm0)
    g = torch.randn_like(out)
    out.backward(g); ref.backward(g)
    torch.testing.assert_close(y.grad, yr.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape,p", [((4, 16, 64, 64), 0.3), ((3, 5, 7, 9), 0.5), ((8, 32, 128, 128), 0.05)])
def test_bn_act_dropout_consistency(shape, p):
    """Dropout draws come from the device RNG: check rate, scaling, and that the backward regenerates the
    same mask (gradient equals autograd through the reference ops with the mask read back from the output)."""
    from uaps_amd import fused, perturb
    torch.manual_seed(3)
    perturb.manual_seed(7)
    B, C, H, W = shape
    y = torch.randn(shape, device=DEV, requires_grad=True)
    bn = nn.BatchNorm2d(C).to(DEV)
    bn_ref = nn.BatchNorm2d(C).to(DEV)
    out = fused.bn_act(y, None, bn, 0.01, p, True)
    yr = y.detach().clone().requires_grad_(True)
    nod = F.leaky_relu(bn_ref(yr), 0.01)
    keep = (out != 0) | (nod == 0)
    rate = float((out != 0).float().mean())
    assert abs(rate - (1 - p)) < 0.01 + 3 * np.sqrt(p * (1 - p) / out.numel())
    ref = nod * keep / (1 - p)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=2e-5)
    g = torch.randn_like(out)
    out.backward(g); ref.backward(g)
    torch.testing.assert_close(y.grad, yr.grad, rtol=1e-3, atol=2e-5 * float(yr.grad.abs().max()))
    torch.testing.assert_close(bn.weight.grad, bn_ref.weight.grad, rtol=1e-3, atol=1e-4 * float(bn_ref.weight.grad.abs().max()))
    # determinism of the stream: same seed -> same mask
    perturb.manual_seed(7)
    out2 = fused.bn_act(y.detach(), None, nn.BatchNorm2d(C).to(DEV), 0.01, p, True)
    assert torch.equal(out2 != 0, out != 0)


@pytest.mark.parametrize("B,Cs,Cl,h,w", [(2, 4, 4, 5, 6), (16, 16, 16, 128, 128), (3, 7, 5, 1, 1), (2, 8, 8, 2, 3), (4, 128, 128, 16, 16),
                                         (1, 2, 3, 40, 36), (2, 1, 2, 70, 64)])
def test_up_cat_vs_torch(B, Cs, Cl, h, w):
    from uaps_amd import fused
    torch.manual_seed(4)
    skip = torch.randn(B, Cs, 2 * h, 2 * w, device=DEV, requires_grad=True)
    low = torch.randn(B, Cl, h, w, device=DEV, requires_grad=True)
    out = fused.up_cat(skip, low)
    sr, lr = skip.detach().clone().requires_grad_(True), low.detach().clone().requires_grad_(True)
    ref = torch.cat([sr, F.interpolate(lr, scale_factor=2, mode="bilinear", align_corners=True)], dim=1)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-6)
    g = torch.randn_like(out)
    out.backward(g); ref.backward(g)
    assert torch.equal(skip.grad, sr.grad)
    torch.testing.assert_close(low.grad, lr.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,Cl,h,w", [(2, 4, 5, 6), (32, 16, 128, 128), (3, 5, 1, 1), (2, 8, 2, 3), (4, 128, 16, 16), (2, 3, 24, 40),
                                      (1, 2, 9, 34), (2, 2, 33, 18), (1, 1, 1, 2), (1, 3, 40, 36), (2, 2, 70, 64)])
def test_upsample2x_vs_torch_and_gather_kernel(B, Cl, h, w):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (UAPS_unet.py:74-75) as the LDS-tiled kernel
    (W % 4 == 0) or the gather kernel, against torch and -- bit for bit -- against the up_cat kernel's interpolation."""
    from uaps_amd import fused
    torch.manual_seed(B * 100 + h)
    low = torch.randn(B, Cl, h, w, device=DEV, requires_grad=True)
    out = fused.upsample2x(low)
    lr = low.detach().clone().requires_grad_(True)
    ref = F.interpolate(lr, scale_factor=2, mode="bilinear", align_corners=True)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-6)
    skip = torch.zeros(B, 1, 2 * h, 2 * w, device=DEV)
    assert torch.equal(out.detach(), fused.up_cat(skip, low.detach())[:, 1:])
    g = torch.randn_like(out)
    out.backward(g); ref.backward(g)
    torch.testing.assert_close(low.grad, lr.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("in_chns,C,n_aux,H,W", [(3, 4, 3, 32, 32), (1, 7, 3, 48, 80), (1, 2, 5, 64, 64)],
                         ids=["neu", "dagm_7class_partial_tiles", "k5_2class"])
def test_forward_pair_equals_two_forwards(in_chns, C, n_aux, H, W, monkeypatch):
    """UNet_UAPS.forward_pair (one pass over labelled+unlabelled, 2 BatchNorm statistics groups) must compute what
    the reference's two forwards compute (UAPS_train.py:177,185): same logits, same running statistics after the
    two successive updates, same loss and parameter gradients.  Randomness is switched off (dropout 0, identity
    perturbations) so the two routes are comparable."""
    import copy
    import uaps_amd
    from uaps_amd import losses, unet
    # plain statistics sums: with the running-mean shift the second of two forwards sees an updated running mean, i.e. another
    # rounding of the same batch statistics, and the two routes agree to a few ulp instead of bit for bit
    monkeypatch.setattr(unet, "_STAT_SHIFT", False)
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    m1 = uaps_amd.UNet_UAPS(in_chns, C, n_aux=n_aux, dropout=(0.0,) * 5).to(dev).train()
    m2 = copy.deepcopy(m1)
    B = 2
    xa, xb = torch.randn(B, in_chns, H, W, device=dev), torch.randn(B, in_chns, H, W, device=dev) * 1.7 + 0.3
    y = torch.randint(0, C, (B, H, W), device=dev)
    ident = [lambda f: f] * n_aux
    w = np.random.default_rng(5).dirichlet(np.ones(n_aux + 1))
    la, lb = m1(xa, ident), m1(xb, ident)
    out1 = losses.uaps_step_loss(la, y, lb, w, 0.07, 0.05)
    out1.loss.backward()
    both = m2.forward_pair(xa, xb, ident)
    out2 = losses.uaps_pair_loss(both, y, w, 0.07, 0.05)
    out2.loss.backward()
    for k in range(n_aux + 1):
        assert torch.equal(both[k][:B], la[k]) and torch.equal(both[k][B:], lb[k]), f"head {k} logits differ"
    assert torch.equal(out1.pseudo, out2.pseudo)
    assert abs(float(out1.loss) - float(out2.loss)) < 1e-6
    for (n1, b1), (n2, b2) in zip(m1.named_buffers(), m2.named_buffers()):
        assert torch.equal(b1, b2), f"buffer {n1} differs"            # running stats: two sequential updates
    worst = 0.0
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        scale = float(p1.grad.abs().max()) + 1e-12
        err = float((p1.grad - p2.grad).abs().max())
        # conv biases in front of a train-mode BatchNorm have an exactly-zero gradient
        assert err <= 5e-5 * scale + 1e-9, f"{n1}: grad err {err:.3e} vs scale {scale:.3e}"
        worst = max(worst, err / scale)


def test_grouped_perturbations_draw_per_group():
    from uaps_amd import perturb
    dev = torch.device("cuda:0")
    perturb.manual_seed(11)
    x = torch.ones(4, 3, 8, 8, device=dev)
    y, noise = perturb.FeatureNoise()(x, return_noise=True, groups=2)
    assert noise.shape == (2, 3, 8, 8) and not torch.equal(noise[0], noise[1])
    assert torch.allclose(y[:2], (1 + noise[0]).expand(2, -1, -1, -1)) and torch.allclose(y[2:], (1 + noise[1]).expand(2, -1, -1, -1))
    assert float(noise.abs().max()) <= 0.3
    z = torch.rand(4, 5, 8, 8, device=dev)
    out, keep = perturb.feature_dropout_with(z, (0.7, 0.9), return_keep=True)
    r0 = perturb.feature_dropout_with(z[:2].contiguous(), 0.7)
    r1 = perturb.feature_dropout_with(z[2:].contiguous(), 0.9)
    assert torch.equal(out[:2], r0) and torch.equal(out[2:], r1)


@pytest.mark.parametrize("B,C,H,W,groups", [(4, 16, 32, 32, 2), (2, 8, 16, 24, 1), (2, 5, 7, 9, 2)])
def test_perturbed_fan_out_backward_equals_separate_kernels(B, C, H, W, groups):
    """perturb.perturbed_fan_out: same forward kernels as FeatureNoise / Dropout / FeatureDropout, and a fused backward
    (one kernel re-applying the three perturbations to the incoming gradients and summing) that must equal the
    per-perturbation backward kernels + fan-in sum; (2,5,7,9) has H*W % 4 != 0 and takes the unfused path."""
    from uaps_amd import perturb
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    f = torch.rand(B, C, H, W, device=dev).requires_grad_(True)
    gs = [torch.randn(B, C, H, W, device=dev) for _ in range(4)]
    kinds = ["noise", "dropout", "feature_dropout"]
    perturb.manual_seed(7); np.random.seed(7)
    outs = perturb.perturbed_fan_out(f, kinds, groups)
    torch.autograd.backward(outs, gs)
    g_fused = f.grad.clone(); f.grad = None
    # reference route: the individual ops with the same RNG state
    perturb.manual_seed(7); np.random.seed(7)
    n_out = perturb.FeatureNoise()(f, groups=groups)
    d_out = perturb.Dropout(f)
    fd_out = perturb.FeatureDropout(f, groups=groups)
    for a, b in zip(outs[1:], (n_out, d_out, fd_out)):
        assert torch.equal(a.detach(), b.detach())
    assert torch.equal(outs[0].detach(), f.detach())
    torch.autograd.backward([f * 1.0, n_out, d_out, fd_out], gs)
    np.testing.assert_allclose(g_fused.cpu().numpy(), f.grad.cpu().numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("B,C,H,W", [(2, 16, 32, 32), (1, 8, 6, 24)])
def test_fused_maxpool_in_fan_out_equals_torch_maxpool(B, C, H, W):
    """perturbed_fan_out(..., with_pool=True): the appended MaxPool2d(2) output and its gradient (routed to the arg-max
    by the fused fan-in kernel, mode 4) must equal torch.nn.functional.max_pool2d and its autograd, ties included."""
    from uaps_amd import perturb
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    f = torch.randn(B, C, H, W, device=dev)
    f[:, :, ::4, ::4] = f[:, :, 1::4, 1::4]                      # plant ties inside pooling windows
    f.requires_grad_(True)
    g_main, g_noise, g_pool = torch.randn_like(f), torch.randn_like(f), torch.randn(B, C, H // 2, W // 2, device=dev)
    perturb.manual_seed(5)
    outs = perturb.perturbed_fan_out(f, ["noise"], 1, with_pool=True)
    assert len(outs) == 3
    ref_pool = F.max_pool2d(f.detach(), 2)
    assert torch.equal(outs[2], ref_pool)
    torch.autograd.backward(outs, [g_main, g_noise, g_pool])
    got = f.grad.clone(); f.grad = None
    perturb.manual_seed(5)
    n_out = perturb.FeatureNoise()(f)
    torch.autograd.backward([f * 1.0, n_out, F.max_pool2d(f, 2)], [g_main, g_noise, g_pool])
    np.testing.assert_allclose(got.cpu().numpy(), f.grad.cpu().numpy(), rtol=1e-6, atol=1e-6)


def test_hip_adam_matches_torch_adam_and_shares_its_state_dict():
    """uaps_amd.optim.Adam (one multi-tensor HIP launch per 48 tensors) against torch.optim.Adam on tensors of assorted
    sizes (odd lengths take the scalar tail), several steps, with weight decay, and a state_dict hand-over mid-run."""
    from uaps_amd import optim
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    shapes = [(16, 3, 3, 3), (16,), (37,), (64, 32, 3, 3), (5, 7), (1,), (256, 256, 3, 3)] + [(8, 8, 3, 3)] * 60
    ref = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    for wd in (0.0, 0.01):
        o_ref = torch.optim.Adam(ref, lr=1e-2, weight_decay=wd)
        o_mine = optim.Adam(mine, lr=1e-2, weight_decay=wd)
        for it in range(5):
            for a, b in zip(ref, mine):
                g = torch.randn_like(a)
                a.grad, b.grad = g, g.clone()
            o_ref.step(); o_mine.step()
            if it == 2:                                            # checkpoint hand-over in both directions
                sd_ref, sd_mine = o_ref.state_dict(), o_mine.state_dict()
                assert sd_ref["state"].keys() == sd_mine["state"].keys() and set(sd_ref["state"][0]) == set(sd_mine["state"][0])
                o_ref.load_state_dict(sd_mine); o_mine.load_state_dict(sd_ref)
        for a, b in zip(ref, mine):
            np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)


@pytest.mark.parametrize("B,Cin,Cout,H,W,ks,groups", [(4, 16, 16, 64, 64, 3, 2), (2, 32, 24, 16, 32, 3, 1), (4, 64, 32, 8, 8, 1, 2),
                                                      (2, 16, 4, 48, 80, 3, 1), (6, 40, 48, 20, 12, 3, 3), (2, 128, 64, 32, 32, 1, 1),
                                                      (4, 32, 32, 256, 256, 3, 2),
                                                      # the full-width-row kernels with the staging-time BatchNorm: 256 wide, and column strips
                                                      (2, 16, 16, 32, 256, 3, 2), (2, 16, 16, 32, 512, 3, 2), (2, 16, 4, 16, 512, 3, 1)])
def test_bn_act_conv_equals_bn_act_then_conv_and_torch(B, Cin, Cout, H, W, ks, groups):
    """conv2(leaky_relu(bn_train(conv1(x)))) with the normalisation applied while conv2 stages its input
    (fused.bn_act_conv: uaps_bn_finalize_train + uaps_conv_fwd_bn + uaps_conv_bwd_weight_partial_bn) against
    (a) the materialising kernels of this package and (b) torch modules on the CPU in fp64, per statistics group."""
    from uaps_amd import conv, fused
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 1000 + Cin * 10 + ks)
    x = torch.randn(B, 8, H, W, generator=g)
    c1 = nn.Conv2d(8, Cin, 3, padding=1)
    c2 = nn.Conv2d(Cin, Cout, ks, padding=ks // 2)
    bn = nn.BatchNorm2d(Cin)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=g); bn.bias.uniform_(-0.5, 0.5, generator=g)
    gout = torch.randn(B, Cout, H, W, generator=g)

    # (b) fp64 CPU reference, one BatchNorm call per group like the reference's two forwards
    c1d, c2d, bnd = (copy_double(m) for m in (c1, c2, bn))
    xd = x.double().requires_grad_(True)
    yd = c1d(xd)
    Bg = B // groups
    ad = torch.cat([F.leaky_relu(bnd(yd[i * Bg:(i + 1) * Bg]), 0.01) for i in range(groups)], 0)
    zd = c2d(ad)
    zd.backward(gout.double())

    def run(fusedpath):
        m1, m2, mb = (copy_to(m, dev) for m in (c1, c2, bn))
        xg = x.to(dev).requires_grad_(True)
        with fused.stat_groups(groups):
            y, st = conv.conv2d_with_stats(xg, m1.weight, None)
            if fusedpath:
                z, zst = fused.bn_act_conv(y, st, m1.bias, mb, 0.01, m2.weight, m2.bias, want_stats=True)
            else:
                a = fused.bn_act(y, m1.bias, mb, 0.01, 0.0, True, st)
                z, zst = conv.conv2d_with_stats(a, m2.weight, m2.bias)
        z.backward(gout.to(dev))
        return (z, zst, xg.grad, m1.weight.grad, m2.weight.grad, m2.bias.grad, mb.weight.grad, mb.bias.grad, mb.running_mean,
                mb.running_var, mb.num_batches_tracked)

    fu, un = run(True), run(False)
    names = ["z", "zstats", "dx", "dw1", "dw2", "db2", "dgamma", "dbeta", "running_mean", "running_var", "nbt"]
    for n_, a_, b_ in zip(names, fu, un):
        scale = float(b_.abs().max()) + 1e-6
        assert float((a_.float() - b_.float()).abs().max()) <= 2e-5 * scale + 1e-6, n_
    refs = [zd, None, xd.grad, c1d.weight.grad, c2d.weight.grad, c2d.bias.grad, bnd.weight.grad, bnd.bias.grad, bnd.running_mean,
            bnd.running_var, bnd.num_batches_tracked]
    for n_, a_, r_ in zip(names, fu, refs):
        if r_ is None:
            continue
        scale = float(r_.abs().max()) + 1e-6
        assert float((a_.cpu().double() - r_.double()).abs().max()) <= 3e-4 * scale + 1e-6, n_


def copy_double(m):
    import copy
    return copy.deepcopy(m).double()


def copy_to(m, dev):
    import copy
    return copy.deepcopy(m).to(dev)


@pytest.mark.parametrize("B,C,H,W,groups,kinds", [(4, 16, 32, 32, 2, ("noise", "dropout", "feature_dropout")),
                                                  (2, 8, 16, 24, 1, ("noise", "dropout", "feature_dropout")),
                                                  (6, 5, 8, 8, 3, ("feature_dropout", "noise")),
                                                  (4, 32, 64, 64, 2, ("noise", "dropout", "feature_dropout", "noise", "dropout"))])
def test_fused_fan_out_forward_equals_the_per_perturbation_kernels(B, C, H, W, groups, kinds):
    """uaps_fanout_perturbed (one pass over f for all perturbed copies, FeatureDropout statistics for the whole batch
    at once) must give bit for bit what the FeatureNoise / Dropout / FeatureDropout kernels give from the same RNG
    state: outputs, the keep mask used by the backward, and the pooled map."""
    from uaps_amd import perturb
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    f = torch.randn(B, C, H, W, device=dev)
    res = []
    for fused_on in (True, False):
        perturb._FUSED_FANOUT = fused_on
        try:
            perturb.manual_seed(21); np.random.seed(21)
            x = f.clone().requires_grad_(True)
            outs = perturb.perturbed_fan_out(x, list(kinds), groups, 0.3, with_pool=True)
            g = [torch.ones_like(o) * (i + 1) for i, o in enumerate(outs)]
            torch.autograd.backward(outs, g)
            res.append(([o.detach() for o in outs], x.grad.clone()))
        finally:
            perturb._FUSED_FANOUT = True
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    assert torch.equal(res[0][1], res[1][1])          # the backward re-applies the same masks / noise


@pytest.mark.parametrize("epilogue", [True, False])
def test_batchnorm_statistics_survive_a_large_channel_mean(epilogue):
    """Train-mode BatchNorm on a conv output whose channel means are thousands of standard deviations: fp32 sums of y and y^2 lose
    the variance to cancellation.  The partial sums (conv epilogue, or the kernel's own statistics pass) are formed about
    running_mean - conv_bias, so a BatchNorm whose running mean tracks the data (any trained checkpoint) normalises correctly."""
    from uaps_amd import conv, fused
    torch.manual_seed(3)
    B, Cc, H, W = 4, 16, 64, 64
    x = (1.0 + 1e-3 * torch.randn(B, Cc, H, W)).to(DEV)
    w = (0.5 + 0.1 * torch.rand(Cc, Cc, 1, 1)).to(DEV)                  # 1x1: no zero-padding border in the statistics
    bn = torch.nn.BatchNorm2d(Cc).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
    y_ref = F.conv2d(x.double().cpu(), w.double().cpu())
    mean, var = y_ref.mean((0, 2, 3)), y_ref.var((0, 2, 3), unbiased=False)
    assert float((mean.abs() / var.sqrt()).min()) > 2e3                 # the regime this test is about
    with torch.no_grad():
        bn.running_mean.copy_(mean.float() * (1 + 1e-4))                 # a running mean that tracks the data, not exactly
    if epilogue:
        y, st = conv.conv2d_with_stats(x, w, None, stat_shift=(bn.running_mean, None))
        out = fused.bn_act(y, None, bn, 1.0, 0.0, True, st)              # slope 1: plain BatchNorm
    else:
        y = conv.conv2d(x, w, None)
        out = fused.bn_act(y, None, bn, 1.0, 0.0, True)
    yd = y.double().cpu()                                               # normalise the kernel's own fp32 y in float64
    m2, v2 = yd.mean((0, 2, 3), keepdim=True), yd.var((0, 2, 3), unbiased=False, keepdim=True)
    ref = (yd - m2) / torch.sqrt(v2 + bn.eps) * bn.weight.double().cpu().view(1, -1, 1, 1) + bn.bias.double().cpu().view(1, -1, 1, 1)
    err = float((out.double().cpu() - ref).abs().max())
    assert err < 5e-3, err


@pytest.mark.parametrize("n_aux", [6, 7])
def test_more_than_five_auxiliary_decoders_run(n_aux):
    """n_aux >= 6 cycles the three perturbations twice, i.e. two FeatureDropout decoders: the one-pass fan-out kernel carries one set
    of thresholds, so these models take the per-perturbation kernels (decided before any random number is reserved).  Forward,
    loss and backward must run and give finite logits and gradients for every head."""
    import uaps_amd
    from uaps_amd import losses
    torch.manual_seed(n_aux)
    m = uaps_amd.UNet_UAPS(3, 4, n_aux=n_aux, feature_chns=[8, 16, 16, 32, 32]).to(DEV).train()
    B, H, W = 2, 32, 32
    xa, xb = torch.randn(B, 3, H, W, device=DEV), torch.randn(B, 3, H, W, device=DEV)
    y = torch.randint(0, 4, (B, H, W), device=DEV)
    both = m.forward_pair(xa, xb)
    assert len(both) == n_aux + 1 and all(z.shape == (2 * B, 4, H, W) and bool(torch.isfinite(z).all()) for z in both)
    w = np.random.default_rng(1).dirichlet(np.ones(n_aux + 1))
    out = losses.uaps_pair_loss(both, y, w, 0.05, 0.05)
    out.loss.backward()
    assert np.isfinite(float(out.loss))
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    assert len({float(z.detach().abs().sum()) for z in both}) == n_aux + 1        # every head sees its own perturbed features


def test_cat_batches_is_torch_cat_with_a_bound():
    """fused.cat_batches (forward_pair's concatenation of the labelled and unlabelled batch): torch.cat bit for bit, and the bound it
    attaches is max|.| of the result; odd sizes and unaligned views take the scalar tail."""
    from uaps_amd import bounds, fused
    dev = torch.device("cuda:0")
    for shape, off in (((16, 3, 64, 64), 0), ((3, 3, 5, 7), 0), ((2, 3, 8, 8), 1)):
        n = int(np.prod(shape))
        g = torch.Generator().manual_seed(n)
        a = (torch.randn(n + off, generator=g) * 3).to(dev)[off:].view(shape)
        b = (torch.randn(n + off, generator=g) * 5).to(dev)[off:].view(shape)
        out = fused.cat_batches(a, b)
        ref = torch.cat([a, b], 0)
        assert torch.equal(out, ref)
        bd = bounds.get(out)
        assert bd is not None and float(bounds.value(bd[0])) * bd[1] == float(ref.abs().max())
    a = torch.randn(2, 3, 8, 8, device=dev, requires_grad=True)
    assert fused.cat_batches(a, a.detach()).requires_grad            # autograd inputs: torch.cat


@pytest.mark.parametrize("groups,n", [(1, 1), (2, 2), (1, 3)])
def test_bn_add_relu_is_relu_of_batchnorm_plus_identity(groups, n):
    """fused.bn_add_relu (the end of a residual block, utilities/resnet.py:85-91): relu(bn_train(y) + identity) in the BatchNorm's
    apply pass, n handles on the result; output, running statistics and every gradient against torch's own modules in float64."""
    from uaps_amd import bounds, conv, fused
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    B, Cin, Cc, H, W = 4, 24, 40, 20, 28
    x = torch.randn(B, Cin, H, W, device=dev)
    w = (torch.randn(Cc, Cin, 1, 1, device=dev) / 5).requires_grad_(True)
    idn = torch.randn(B, Cc, H, W, device=dev, requires_grad=True)
    bn = nn.BatchNorm2d(Cc).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
    ref_bn = nn.BatchNorm2d(Cc).double()
    ref_bn.load_state_dict({k: v.detach().cpu().double() if v.is_floating_point() else v.cpu() for k, v in bn.state_dict().items()})
    with fused.stat_groups(groups):
        y, st = conv.conv2d_with_stats(x, w, None)
        outs = fused.bn_add_relu(y, st, bn, idn, n)
    outs = (outs,) if n == 1 else outs
    gs = [torch.randn(B, Cc, H, W, device=dev) for _ in range(n)]
    sum((o * g).sum() for o, g in zip(outs, gs)).backward()
    assert all(torch.equal(o, outs[0]) for o in outs)
    bd = bounds.get(outs[0])
    if bd is not None:
        assert float(bounds.value(bd[0])) * bd[1] == float(outs[0].max())
    # torch, float64, per statistics group
    xr, wr, ir = x.cpu().double(), w.detach().cpu().double().requires_grad_(True), idn.detach().cpu().double().requires_grad_(True)
    yr = F.conv2d(xr, wr)
    Bg = B // groups
    parts = [ref_bn(yr[g * Bg:(g + 1) * Bg]) for g in range(groups)]
    outr = torch.relu(torch.cat(parts, 0) + ir)
    (outr * sum(g.cpu().double() for g in gs)).sum().backward()
    assert float((outs[0].detach().cpu().double() - outr.detach()).abs().max()) < 1e-4
    assert float((idn.grad.cpu().double() - ir.grad).abs().max()) < 1e-4
    assert float((w.grad.cpu().double() - wr.grad).abs().max()) < 1e-3
    assert float((bn.weight.grad.cpu().double() - ref_bn.weight.grad).abs().max()) < 1e-3
    assert float((bn.bias.grad.cpu().double() - ref_bn.bias.grad).abs().max()) < 1e-3
    assert torch.allclose(bn.running_mean.cpu().double(), ref_bn.running_mean, atol=1e-5)
    assert torch.allclose(bn.running_var.cpu().double(), ref_bn.running_var, rtol=1e-4)
