"""GPU parity of the hand-written convolution kernels (csrc/conv_kernels.hpp) against plain PyTorch
fp32 convolutions evaluated on the CPU (the arithmetic the reference's nn.Conv2d layers perform,
utilities/UAPS_unet.py:37,41,73,138).  Tolerance: fp32 sums of K = 9*Cin (forward), 9*Cout (input
gradient) or B*H*W (weight gradient) products in a different order -> 2e-5 of the output scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["h16", "split", "exact"])
def conv_mode(request):
    """Every test of this file runs in the three arithmetic modes of the convolution kernels: two fp16 pieces per scaled operand
    (the default; needs the operands' magnitude bounds, which _b() attaches), the three-way bf16 split, and the fp32 matrix
    instruction."""
    from uaps_amd import conv
    prev = conv.get_mode()
    conv.set_mode(request.param)
    yield request.param
    conv.set_mode(prev)


def _b(t, slack=1.0):
    """Attach the tensor's magnitude bound (uaps_amd/bounds.py) the way the producing kernels do: a device scalar >= max|t|."""
    from uaps_amd import bounds
    return bounds.put(t, bounds.from_value(t.detach().abs().max() * slack), 1.0)

# (B, Cin, Cout, H, W, ks)
SHAPES = [
    (2, 3, 16, 32, 32, 3),      # first encoder conv (Cin padded to 4)
    (2, 16, 16, 64, 64, 3),
    (1, 32, 16, 40, 72, 3),     # concat conv, partial tiles in both directions
    (2, 16, 4, 32, 32, 3),      # out_conv, C=4
    (2, 16, 4, 64, 128, 3),     # out_conv on a wide map: the exact-N VALU kernels (csrc/conv_small.hpp) in all three directions
    (1, 16, 2, 40, 72, 3),      # ... C=2, partial tiles
    (2, 3, 16, 24, 64, 3),      # <= 4 contraction channels on a wide map
    (1, 16, 7, 16, 48, 3),      # out_conv, DAGM C=7
    (1, 16, 2, 24, 20, 3),      # narrow map (16x16 tile path), C=2
    (2, 64, 64, 16, 16, 3),
    (1, 128, 256, 16, 16, 3),
    (2, 256, 128, 16, 16, 1),   # UpBlock conv1x1
    (4, 256, 128, 32, 32, 1),   # ... of up1 at a 512 x 512 input: 1024 pixels, the GEMM-tiled plan (plan_fwd's big_1x1 rule)
    (2, 32, 16, 64, 64, 1),
    (1, 20, 40, 18, 18, 3),     # channel counts that are not multiples of the tile sizes
    (1, 24, 24, 10, 10, 1),
    # wide 1x1 projections of the ResNet bottlenecks (utilities/resnet.py:55-95): the GEMM-tiled kernels of csrc/conv_gemm1x1.hpp
    (2, 256, 64, 40, 40, 1),    # 1600 pixels: a partial 128-pixel tile; 64-channel output blocks
    (2, 64, 256, 32, 32, 1),    # 128-channel output blocks, one chunk pair
    (1, 512, 128, 32, 48, 1),
    (2, 200, 136, 32, 32, 1),   # channel counts that are no multiples of the block sizes (stays on the 3x3-style tiling: CoutP % 64)
    (1, 1024, 256, 40, 40, 1),
    (1, 128, 512, 24, 24, 1),   # two 256-channel output blocks (conv_g1h256_kernel), a partial pixel tile
    # 32-channel blocks on 16-row tiles (csrc/conv_split.hpp, MR = 4): >= 512 tiles of 16 x 32 pixels, ragged in both directions
    (5, 24, 32, 200, 264, 3),
    (4, 32, 32, 256, 256, 3),
    # 256-wide maps, <= 16 output channels: the full-width-row kernels (csrc/conv_split_row16.hpp forward / input gradient,
    # csrc/conv_split_wrw_row.hpp weight gradient) in the fp16 form; runs of 16 rows, several runs per workgroup at B = 40
    (2, 16, 16, 32, 256, 3),
    (3, 32, 16, 48, 256, 3),
    (2, 16, 4, 32, 256, 3),     # out_conv: 4 classes on one padded 16-channel tile
    (2, 32, 12, 16, 256, 3),
    (3, 3, 16, 32, 256, 3),     # the first layer: its weight gradient rides in the 16-channel row kernel, absent channels never fetched
    (2, 20, 12, 16, 256, 3),    # ... and 20 of 32 channels
    # wider than 256 pixels: the same kernels on 256-wide column strips, one real neighbour pixel either side of a strip
    (2, 16, 16, 32, 512, 3),
    (1, 32, 16, 16, 768, 3),
    (2, 16, 4, 16, 512, 3),
    (2, 3, 16, 16, 512, 3),
    (2, 32, 16, 32, 256, 3),    # its input gradient (16 -> 32 channels) takes two output tiles of the row kernel
    (40, 16, 16, 256, 256, 3),
]


def _mk(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32)


@pytest.mark.parametrize("B,Cin,Cout,H,W,ks", SHAPES)
def test_conv_forward_backward_vs_torch_cpu(B, Cin, Cout, H, W, ks):
    from uaps_amd.conv import conv2d
    x = _mk((B, Cin, H, W), 1)
    w = _mk((Cout, Cin, ks, ks), 2) / np.sqrt(Cin * ks * ks)
    b = _mk((Cout,), 3)
    dy = _mk((B, Cout, H, W), 4)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=ks // 2)
    yr.backward(dy)
    dev = torch.device("cuda:0")
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y = conv2d(_b(xg), wg, bg)
    y.backward(_b(dy.to(dev)))

    def close(a, ref, what):
        scale = float(ref.abs().max()) + 1e-12
        err = float((a.cpu() - ref).abs().max())
        assert err <= 2e-5 * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"

    close(y.detach(), yr.detach(), "y")
    close(xg.grad, xr.grad, "dx")
    close(wg.grad, wr.grad, "dw")
    close(bg.grad, br.grad, "db")


def test_conv_no_bias_and_cached_pack_follows_inplace_updates():
    from uaps_amd.conv import conv2d
    dev = torch.device("cuda:0")
    x = _mk((1, 8, 16, 16), 5).to(dev)
    w = torch.nn.Parameter(_mk((8, 8, 3, 3), 6).to(dev))
    y0 = conv2d(x, w)
    assert torch.allclose(y0.cpu(), F.conv2d(x.cpu(), w.detach().cpu(), padding=1), atol=1e-4)
    with torch.no_grad():
        w.mul_(2.0)                      # what optimizer.step() does: in-place update -> version bump -> repack
    y1 = conv2d(x, w)
    assert torch.allclose(y1, 2 * y0, rtol=1e-6, atol=1e-6)


def test_packed_weights_follow_optimizer_steps():
    """Fused Adam updates parameters without bumping Tensor._version: the packed-weight cache must
    still notice (global optimizer post-step hook), otherwise training silently uses stale weights."""
    from uaps_amd.conv import conv2d
    dev = torch.device("cuda:0")
    x = _mk((1, 8, 16, 16), 5).to(dev)
    w = torch.nn.Parameter(_mk((8, 8, 3, 3), 6).to(dev))
    for fused in (True, False):
        opt = torch.optim.Adam([w], lr=0.1, fused=fused)
        conv2d(x, w).square().mean().backward()
        opt.step()
        y = conv2d(x, w)
        ref = F.conv2d(x.cpu(), w.detach().cpu(), padding=1)
        assert torch.allclose(y.detach().cpu(), ref, atol=1e-4), f"stale packed weights after Adam(fused={fused})"
        opt.zero_grad()


def test_conv_is_deterministic():
    from uaps_amd.conv import conv2d
    dev = torch.device("cuda:0")
    x = _mk((2, 32, 32, 32), 7).to(dev).requires_grad_(True)
    w = _mk((32, 32, 3, 3), 8).to(dev).requires_grad_(True)
    dy = _mk((2, 32, 32, 32), 9).to(dev)
    outs = []
    for _ in range(2):
        x.grad = w.grad = None
        y = conv2d(x, w)
        y.backward(dy)
        outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)          # no float atomics anywhere: bitwise reproducible


def test_conv_refuses_cpu_tensors():
    from uaps_amd.conv import conv2d
    from uaps_amd._lib import UapsHipError
    with pytest.raises(UapsHipError):
        conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3))


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(4, 16, 16, 64, 64), (2, 8, 40, 24, 36), (2, 32, 128, 16, 16),
                                            (4, 32, 32, 256, 256), (4, 16, 64, 252, 256),           # these two: persistent slice kernels
                                            (6, 24, 32, 200, 264),          # 32-channel block on 16-row tiles, the last one half outside: parts stay per 8 rows
                                            (4, 16, 16, 64, 256), (4, 32, 16, 32, 256), (2, 16, 10, 48, 256),      # full-width-row kernels: parts stay 8 x 32 tiles
                                            (2, 16, 16, 32, 512), (2, 32, 16, 16, 768)])     # ... on column strips
def test_conv_epilogue_statistics_feed_batchnorm(B, Cin, Cout, H, W):
    """conv2d_with_stats: the per-tile (sum, sum of squares) written by the conv epilogue must add up to the
    statistics of y, and bn_act fed with them must equal bn_act running its own statistics pass."""
    from uaps_amd import fused
    from uaps_amd.conv import conv2d, conv2d_with_stats
    dev = torch.device("cuda:0")
    x = _b(_mk((B, Cin, H, W), 11).to(dev))
    w = (_mk((Cout, Cin, 3, 3), 12) / np.sqrt(9 * Cin)).to(dev)
    y, st = conv2d_with_stats(x, w)
    assert torch.equal(y, conv2d(x, w))
    assert st.shape[:2] == (Cout, B) and st.shape[-1] == 2
    s_ref = y.double().sum(dim=(2, 3)).t()                       # [Cout, B]
    q_ref = (y.double() ** 2).sum(dim=(2, 3)).t()
    np.testing.assert_allclose(st[..., 0].double().sum(-1).cpu().numpy(), s_ref.cpu().numpy(), rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(st[..., 1].double().sum(-1).cpu().numpy(), q_ref.cpu().numpy(), rtol=1e-5)
    for groups in (1, 2):
        bn1, bn2 = torch.nn.BatchNorm2d(Cout).to(dev), torch.nn.BatchNorm2d(Cout).to(dev)
        with fused.stat_groups(groups):
            a1 = fused.bn_act(y, None, bn1, 0.01, 0.0, True)
            a2 = fused.bn_act(y, None, bn2, 0.01, 0.0, True, st)
        assert float((a1 - a2).abs().max()) < 2e-5
        assert torch.allclose(bn1.running_mean, bn2.running_mean, atol=1e-6) and torch.allclose(bn1.running_var, bn2.running_var, rtol=1e-5)
        assert int(bn2.num_batches_tracked) == groups


@pytest.mark.parametrize("B,C1,C2,Cout,H,W,ks", [(2, 16, 16, 16, 64, 64, 3), (2, 32, 32, 32, 32, 32, 3), (1, 128, 128, 128, 16, 16, 3),
                                                  (2, 16, 24, 20, 24, 40, 3), (2, 32, 16, 8, 16, 16, 1), (4, 16, 16, 32, 256, 256, 3),
                                                  (3, 16, 16, 16, 32, 256, 3),       # two tensors into the full-width-row kernels
                                                  (2, 16, 16, 16, 16, 512, 3), (2, 16, 16, 32, 32, 512, 3)])      # ... on column strips (the second: 16 + 16 output channels)
def test_conv_cat_equals_conv_of_concatenation(B, C1, C2, Cout, H, W, ks):
    """conv2d_cat(x1, x2, w) must be conv2d(cat([x1, x2]), w) bit for bit (same kernels, same order of operations),
    and so must its three gradients."""
    from uaps_amd.conv import conv2d, conv2d_cat
    dev = torch.device("cuda:0")
    x1, x2 = _mk((B, C1, H, W), 21).to(dev).requires_grad_(True), _mk((B, C2, H, W), 22).to(dev).requires_grad_(True)
    w = (_mk((Cout, C1 + C2, ks, ks), 23) / np.sqrt((C1 + C2) * ks * ks)).to(dev).requires_grad_(True)
    b = _mk((Cout,), 24).to(dev).requires_grad_(True)
    dy = _b(_mk((B, Cout, H, W), 25).to(dev))
    y, st = conv2d_cat(_b(x1), _b(x2), w, b, with_stats=True)
    y.backward(dy)
    g = (x1.grad.clone(), x2.grad.clone(), w.grad.clone(), b.grad.clone())
    x1.grad = x2.grad = w.grad = b.grad = None
    xc = _b(torch.cat([x1, x2], dim=1).detach().requires_grad_(True))
    yr = conv2d(xc, w, b)
    yr.backward(dy)
    assert torch.equal(y, yr)
    assert torch.equal(g[0], xc.grad[:, :C1]) and torch.equal(g[1], xc.grad[:, C1:])
    assert torch.equal(g[2], w.grad) and torch.equal(g[3], b.grad)
    np.testing.assert_allclose(st[..., 0].double().sum((1, 2)).cpu().numpy(), y.detach().double().sum((0, 2, 3)).cpu().numpy(), rtol=1e-4, atol=1e-2)


def test_wide_1x1_over_two_tensors_takes_the_exact_plan_in_every_entry_point():
    """The GEMM-tiled 1x1 plan has no two-tensor form and its own statistics / workspace layouts: conv.plan_cfg gives such a call
    cfg bit 28 in all its entry points (statistics parts, forward, both gradients, workspace, reduce), so they agree -- instead of
    UAPS_ERANGE from the library or statistics in a layout the consumer does not expect."""
    from uaps_amd import conv
    from uaps_amd.conv import conv2d_cat
    dev = torch.device("cuda:0")
    B, C1, C2, Cout, H, W = 2, 64, 64, 128, 32, 32
    if conv.get_mode() != "exact":
        assert conv.kernel_variant("fwd", B, C1 + C2, Cout, H, W, 1).startswith("conv_g1")      # what the single-tensor layer runs
    assert conv.plan_cfg(1, 0, False) == 1 << 28 and conv.plan_cfg(3, 0, False) == 0
    x1, x2 = _mk((B, C1, H, W), 31).to(dev).requires_grad_(True), _mk((B, C2, H, W), 32).to(dev).requires_grad_(True)
    w = (_mk((Cout, C1 + C2, 1, 1), 33) / np.sqrt(C1 + C2)).to(dev).requires_grad_(True)
    dy = _b(_mk((B, Cout, H, W), 35).to(dev))
    y, st = conv2d_cat(_b(x1), _b(x2), w, None, with_stats=True)
    y.backward(dy)
    xc = torch.cat([x1, x2], 1).detach().double().cpu().requires_grad_(True)
    wc = w.detach().double().cpu().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xc, wc)
    yr.backward(dy.double().cpu())
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 1e-4
    assert float((torch.cat([x1.grad, x2.grad], 1).cpu().double() - xc.grad).abs().max()) < 1e-4
    assert float((w.grad.cpu().double() - wc.grad).abs().max()) < 1e-3 * float(wc.grad.abs().max())
    np.testing.assert_allclose(st[..., 0].double().sum((1, 2)).cpu().numpy(), yr.detach().sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(st[..., 1].double().sum((1, 2)).cpu().numpy(), (yr.detach() ** 2).sum((0, 2, 3)).numpy(), rtol=1e-4)


def test_packed_buffers_of_a_transient_weight_die_with_it():
    """conv3x3s2 builds its re-arranged kernel as a fresh tensor on every forward: the packed-weight cache must not keep an entry
    (2 x 8 MB for layer2.0.conv2) per call until some later purge."""
    import gc
    from uaps_amd import conv
    dev = torch.device("cuda:0")
    x = _b(_mk((1, 16, 16, 32), 41).to(dev))
    gc.collect()
    n0 = len(conv._packed)
    for i in range(20):
        w = (_mk((16, 16, 3, 3), 50 + i) / 12).to(dev)
        conv.conv2d(x, w, None)
        del w
    gc.collect()
    assert len(conv._packed) <= n0 + 1


@pytest.mark.parametrize("B,Cin,Cout,H,W,dil", [(2, 32, 32, 40, 40, 2), (1, 64, 48, 32, 32, 4), (2, 16, 128, 20, 36, 2), (1, 256, 256, 16, 16, 4),
                                                (1, 128, 128, 80, 80, 2), (1, 256, 256, 80, 80, 4), (2, 64, 96, 12, 44, 4), (1, 8, 32, 24, 24, 2)])
def test_dilated_conv_vs_torch_cpu(B, Cin, Cout, H, W, dil):
    """3x3 convolutions with dilation 2 / 4 and padding = dilation (the stride-replaced-by-dilation stages of the
    reference's ResNet-50, utilities/resnet.py:8-10, 201-203): forward, input and weight gradient vs PyTorch CPU.  With
    >= 32 output channels forward and input gradient run the dilated split kernels (csrc/conv_split.hpp, DIL = 2 / 4)."""
    from uaps_amd.conv import conv2d
    x = _mk((B, Cin, H, W), 31)
    w = _mk((Cout, Cin, 3, 3), 32) / np.sqrt(Cin * 9)
    dy = _mk((B, Cout, H, W), 33)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, padding=dil, dilation=dil)
    yr.backward(dy)
    dev = torch.device("cuda:0")
    xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    y = conv2d(_b(xg), wg, None, dilation=dil)
    y.backward(_b(dy.to(dev)))
    for got, ref, what in ((y.detach(), yr.detach(), "y"), (xg.grad, xr.grad, "dx"), (wg.grad, wr.grad, "dw")):
        scale = float(ref.abs().max()) + 1e-12
        err = float((got.cpu() - ref).abs().max())
        assert err <= 2e-5 * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("B,Cin,Cout,H,W,ks", [(4, 64, 64, 64, 64, 3), (2, 16, 16, 256, 256, 3), (4, 256, 128, 16, 16, 1), (2, 3, 16, 128, 128, 3),
                                               (2, 128, 128, 32, 32, 3), (4, 32, 32, 256, 256, 3), (4, 16, 64, 256, 256, 3)])
def test_split_mode_is_as_accurate_as_the_fp32_matrix_instruction(B, Cin, Cout, H, W, ks, conv_mode):
    """The bf16-split kernels claim fp32 accuracy: measured against a float64 reference, their forward / input-gradient
    error must be of the size of the exact fp32 kernels' (both are a few 1e-7 of sum |a*b|), on inputs with a wide
    dynamic range (log-normal magnitudes) so that all three pieces of every operand matter."""
    if conv_mode != "split":
        pytest.skip("compares the two modes itself")
    from uaps_amd import conv
    g = torch.Generator().manual_seed(B * 7 + Cin + ks)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(2.0 * torch.randn(B, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * torch.exp(torch.randn(Cout, Cin, ks, ks, generator=g)) / np.sqrt(Cin * ks * ks)
    dy = torch.randn(B, Cout, H, W, generator=g) * torch.exp(2.0 * torch.randn(B, Cout, H, W, generator=g))
    xr, wr = x.double().requires_grad_(True), w.double()
    yr = F.conv2d(xr, wr, None, padding=ks // 2)
    yr.backward(dy.double())
    mag_y = F.conv2d(x.double().abs(), wr.abs(), None, padding=ks // 2)                     # sum |a*b| per output
    mag_dx = torch.nn.grad.conv2d_input(x.shape, wr.abs(), dy.double().abs(), padding=ks // 2)
    dev = torch.device("cuda:0")
    errs = {}
    for mode in ("h16", "split", "exact"):
        conv.set_mode(mode)
        xg, wg = x.to(dev).requires_grad_(True), w.to(dev)
        y = conv.conv2d(_b(xg), wg)
        y.backward(_b(dy.to(dev)))
        errs[mode] = (float(((y.detach().cpu().double() - yr.detach()).abs() / mag_y).max()),
                      float(((xg.grad.cpu().double() - xr.grad).abs() / mag_dx).max()))
    conv.set_mode("split")
    print(errs)
    for i, what in enumerate(("y", "dx")):
        for mode in ("h16", "split"):
            assert errs[mode][i] < 1.5 * errs["exact"][i] + 1e-7, (mode, what, errs)
            assert errs[mode][i] < 5e-6, (mode, what, errs)                # all are a few fp32 roundings of sum |a*b| (K up to 2304)


@pytest.mark.parametrize("xs,ws,dys", [(1e-30, 1.0, 1e30), (1e30, 1e-6, 1e-30), (1.0, 1e-30, 1.0), (3e4, 50.0, 1e-9)])
@pytest.mark.parametrize("Cin,Cout,H,W", [(32, 64, 32, 32), (16, 16, 64, 64)])
def test_fp16_pieces_hold_any_fp32_range(xs, ws, dys, Cin, Cout, H, W, conv_mode):
    """The two-piece fp16 form scales every operand by a power of two taken from its bound: tensors of any fp32 magnitude
    (1e-30 ... 1e30, far outside fp16's 6e-8 ... 65504), a bound that is 1000x too large, and an all-zero operand give the
    accuracy of the fp32 kernels -- relative to sum |a b|, against float64."""
    if conv_mode != "h16":
        pytest.skip("h16 only")
    from uaps_amd import conv
    g = torch.Generator().manual_seed(Cin + H)
    dev = torch.device("cuda:0")
    x = torch.randn(2, Cin, H, W, generator=g) * torch.exp(torch.randn(2, Cin, H, W, generator=g)) * xs
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(9 * Cin) * ws
    dy = torch.randn(2, Cout, H, W, generator=g) * torch.exp(torch.randn(2, Cout, H, W, generator=g)) * dys
    for slack in (1.0, 1000.0):
        conv.KERNEL_EVENTS, conv.EVENT_FILTER = {}, None
        xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
        y = conv.conv2d(_b(xg, slack), wg)
        y.backward(_b(dy.to(dev), slack))
        names, conv.KERNEL_EVENTS = set(conv.KERNEL_EVENTS), None
        assert any(n.startswith(("conv_h32", "conv_hfwd", "conv_hp16")) for n in names) and any(n.startswith("conv_hwrw") for n in names), names
        xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
        yr = F.conv2d(xr, wr, None, padding=1)
        yr.backward(dy.double())
        mag_y = F.conv2d(x.double().abs(), w.double().abs(), None, padding=1)
        mag_dx = torch.nn.grad.conv2d_input(x.shape, w.double().abs(), dy.double().abs(), padding=1)
        mag_dw = torch.nn.grad.conv2d_weight(x.double().abs(), w.shape, dy.double().abs(), padding=1)
        for got, ref, mag, what in ((y.detach(), yr.detach(), mag_y, "y"), (xg.grad, xr.grad, mag_dx, "dx"), (wg.grad, wr.grad, mag_dw, "dw")):
            assert torch.isfinite(got).all(), what
            err = float(((got.cpu().double() - ref).abs() / mag).max())
            assert err < 5e-6, (what, slack, err)
    # an all-zero operand (bound 0: no scaling) and a zero gradient
    xg, wg = torch.zeros(2, Cin, H, W, device=dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    y = conv.conv2d(_b(xg), wg)
    y.backward(_b(torch.zeros_like(y)))
    assert float(y.abs().max()) == 0.0 and float(xg.grad.abs().max()) == 0.0 and float(wg.grad.abs().max()) == 0.0


def test_conv_without_bounds_runs_the_bf16_form_in_h16_mode(conv_mode):
    if conv_mode != "h16":
        pytest.skip("h16 only")
    from uaps_amd import conv
    dev = torch.device("cuda:0")
    x, w = _mk((2, 32, 32, 32), 5).to(dev).requires_grad_(True), (_mk((32, 32, 3, 3), 6) / 17).to(dev).requires_grad_(True)
    conv.KERNEL_EVENTS, conv.EVENT_FILTER = {}, None
    y = conv.conv2d(x, w)
    y.backward(torch.ones_like(y))
    names, conv.KERNEL_EVENTS = set(conv.KERNEL_EVENTS), None
    assert names and all(n.startswith("conv_s") for n in names), names


def test_launch_timer_reads_the_kernel_dispatch(conv_mode):
    """_lib.LaunchTimer (uaps_next_launch_events): the events ride on the convolution kernel's dispatch, so the elapsed time
    is positive, not more than what two events bracketing the call on the stream measure (up to timestamp granularity), and
    the results are untouched."""
    from uaps_amd import _lib, conv
    from uaps_amd.conv import conv2d
    dev = torch.device("cuda:0")
    x = _b(_mk((8, 32, 128, 128), 21).to(dev))
    w = _mk((32, 32, 3, 3), 22).to(dev)
    ref = conv2d(x, w)
    torch.cuda.synchronize()
    for kind in ("fwd", "bwd_data", "wrw"):
        conv.KERNEL_EVENTS, conv.EVENT_FILTER = {}, None
        try:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            xr = x.detach().clone().requires_grad_(kind == "bwd_data")
            xr = _b(xr)
            wr = w.detach().clone().requires_grad_(kind == "wrw")
            y = conv2d(xr, wr)
            dy = _b(torch.ones_like(y))
            s.record()
            y.backward(dy) if kind != "fwd" else conv2d(x, w)
            e.record()
            torch.cuda.synchronize()
            recs = {k: v for k, v in conv.KERNEL_EVENTS.items()}
        finally:
            conv.KERNEL_EVENTS = None
        assert torch.equal(y.detach(), ref)
        inner = sum(a.elapsed_time(b) for v in recs.values() for a, b, *_ in v)
        # (dispatch timestamps and event records are taken by different agents: allow their granularity)
        assert 0.0 < inner <= 1.25 * s.elapsed_time(e) + 0.01, (kind, inner, s.elapsed_time(e))
    with _lib.LaunchTimer() as t:           # nothing launched inside: the pair falls back to bracketing the (empty) block
        pass
    torch.cuda.synchronize()
    assert not t.dispatch and t.elapsed_ms() >= 0.0


def test_the_explicit_conv_call_equals_the_hinted_entry_points():
    """uaps_conv_ex (one size-versioned struct per call, include/uaps_hip.h) against the *_h entry points the package calls and
    against the legacy pair uaps_next_call_hints + uaps_conv_fwd: the same kernels, bit-identical results; a pending thread-local
    record is neither used nor consumed by the explicit forms; an old client's shorter struct is accepted."""
    import ctypes as C
    from uaps_amd import _lib, bounds, conv
    DEV = "cuda:0"
    if conv.get_mode() != "h16":
        pytest.skip("one arithmetic mode is enough for the ABI form")
    torch.manual_seed(4)
    B, Cin, Cout, H, W, ks = 2, 32, 32, 64, 64, 3
    x = torch.randn(B, Cin, H, W, device=DEV)
    dy = torch.randn(B, Cout, H, W, device=DEV)
    w = torch.randn(Cout, Cin, ks, ks, device=DEV) * 0.1
    wf, wb = conv.pack_weights(w)
    xb, dyb = (bounds.from_value(x.abs().max()), 1.0), (bounds.from_value(dy.abs().max()), 1.0)
    y_ref = conv.conv_fwd_raw(x, wf, None, Cout, ks, xb=xb)
    dx_ref = conv.conv_bwd_data_raw(dy, wb, Cin, ks, dyb=dyb)
    L = _lib.lib()
    st = _lib.current_stream(x.device)

    def call(op, **kw):
        c = _lib.ConvCall()
        c.struct_size = C.sizeof(_lib.ConvCall)
        c.op, c.B, c.Cin, c.Cout, c.H, c.W, c.ks, c.cfg, c.C1, c.stream = op, B, Cin, Cout, H, W, ks, 0, Cin, st
        for k, v in kw.items():
            setattr(c, k, v)
        return c

    def with_bound(c, *bs):
        c.hints.struct_size = C.sizeof(_lib.CallHints)
        for i, b in enumerate(bs):
            c.hints.bound[i], c.hints.mul[i] = b[0].data_ptr(), b[1]
        return c

    y = torch.empty_like(y_ref)
    # a pending record of the legacy protocol with a bound that is far too small (fp16 overflow if anything used it): the explicit
    # forms must neither use nor consume it
    tiny = (bounds.from_value(torch.tensor(1e-20, device=DEV)), 1.0)
    pending = _lib.CallHints()
    pending.struct_size = C.sizeof(_lib.CallHints)
    pending.bound[0], pending.mul[0] = tiny[0].data_ptr(), 1.0
    assert L.uaps_next_call_hints(C.byref(pending)) == 0
    with _lib.device_guard(x.device):
        rc = L.uaps_conv_ex(C.byref(with_bound(call(0, x=x.data_ptr(), w_packed=wf.data_ptr(), y=y.data_ptr()), xb)))
    _lib.check(rc, "uaps_conv_ex fwd")
    assert torch.equal(y, y_ref)
    y.zero_()
    with _lib.device_guard(x.device):        # the *_h form with the record as its first argument
        _lib.check(L.uaps_conv_fwd_h(_lib.mk_hints((xb,)), x.data_ptr(), wf.data_ptr(), None, y.data_ptr(), B, Cin, Cout, H, W, ks, 0, st), "uaps_conv_fwd_h")
    assert torch.equal(y, y_ref)
    # ... and the pending record is still there for the legacy entry point that follows: replace it with the true bound and compare
    assert L.uaps_next_call_hints(None) == 0
    good = _lib.CallHints()
    good.struct_size = C.sizeof(_lib.CallHints)
    good.bound[0], good.mul[0] = xb[0].data_ptr(), xb[1]
    y.zero_()
    with _lib.device_guard(x.device):
        assert L.uaps_next_call_hints(C.byref(good)) == 0
        _lib.check(L.uaps_conv_fwd(x.data_ptr(), wf.data_ptr(), None, y.data_ptr(), B, Cin, Cout, H, W, ks, 0, st), "uaps_conv_fwd (legacy hints)")
    assert torch.equal(y, y_ref)
    # a hints record that claims more bytes than the caller's struct leaves for it
    c = with_bound(call(0, x=x.data_ptr(), w_packed=wf.data_ptr(), y=y.data_ptr()), xb)
    c.struct_size = _lib.ConvCall.hints.offset + 32
    assert L.uaps_conv_ex(C.byref(c)) == -1
    dx = torch.empty_like(dx_ref)
    with _lib.device_guard(x.device):
        rc = L.uaps_conv_ex(C.byref(with_bound(call(1, x=dy.data_ptr(), w_packed=wb.data_ptr(), y=dx.data_ptr()), dyb)))
    _lib.check(rc, "uaps_conv_ex bwd_data")
    assert torch.equal(dx, dx_ref)
    # weight gradient partials + the classic reduce, against the classic pair
    dw_ref, _ = conv.conv_bwd_weight_raw(dy, x, ks, False, dyb=dyb, xb=xb)
    n = C.c_size_t()
    _lib.check(L.uaps_conv_wrw_workspace_bytes(B, Cin, Cout, H, W, ks, 0, C.byref(n)), "ws")
    ws = torch.empty(n.value // 4 + 16, dtype=torch.float32, device=DEV)
    dw = torch.empty_like(w)
    with _lib.device_guard(x.device):
        c = with_bound(call(2, x=x.data_ptr(), y_grad=dy.data_ptr(), workspace=ws.data_ptr(), workspace_bytes=ws.numel() * 4), dyb, xb)
        _lib.check(L.uaps_conv_ex(C.byref(c)), "uaps_conv_ex bwd_weight")
        _lib.check(L.uaps_conv_bwd_weight_reduce(ws.data_ptr(), dw.data_ptr(), None, B, Cin, Cout, H, W, ks, 0, st), "reduce")
    assert torch.equal(dw, dw_ref)
    # an older client's struct that ends in front of `hints`: accepted, runs without bounds (the exact three-piece form)
    y2 = torch.empty_like(y_ref)
    c = call(0, x=x.data_ptr(), w_packed=wf.data_ptr(), y=y2.data_ptr())
    c.struct_size = _lib.ConvCall.hints.offset
    with _lib.device_guard(x.device):
        _lib.check(L.uaps_conv_ex(C.byref(c)), "uaps_conv_ex short struct")
    assert float((y2 - y_ref).abs().max()) <= 2e-5 * float(y_ref.abs().max())
    c.struct_size = 8
    assert L.uaps_conv_ex(C.byref(c)) == -1 and L.uaps_conv_ex(None) == -1


@pytest.mark.parametrize("lazy", [False, True])
def test_up_sampling_in_the_staging_equals_the_materialised_operand(lazy):
    """conv2d_cat(skip, low, w, up2=True) -- up4's first convolution reading the 1x1 projection's LOW-resolution output and
    up-sampling it x2 while staging (UAPS_unet.py:74-75, 83-85; csrc/up2_staging.hpp) -- against the same convolution on the
    materialised nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) output with the same operand bound: forward,
    BatchNorm partial sums and every gradient bit for bit (the staged values are identical, the kernels' summation orders too);
    the up-sampled operand itself against torch.  lazy: the weight gradient also applies a pending BatchNorm transform (the
    form the training step runs)."""
    import torch.nn as nn
    from uaps_amd import bounds, conv, fused, lazybn
    if conv.get_mode() != "h16":
        pytest.skip("the up-sampling forms exist in the fp16-split arithmetic only")
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    B, H, W = 4, 32, 256
    skip0 = torch.randn(B, 16, H, W, device=dev)
    low0 = torch.randn(B, 16, H // 2, W // 2, device=dev) * 1.7
    w1 = (torch.randn(16, 32, 3, 3, device=dev) / 10)
    w2 = (torch.randn(16, 16, 3, 3, device=dev) / 10)
    bn = nn.BatchNorm2d(16).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
    bounds.refresh([bn])
    g = torch.randn(B, 16, H, W, device=dev)
    assert conv.up2_eligible(bounds.put(skip0.clone(), bounds.from_value(skip0.abs().max())), w1)

    def run(fusedp):
        skip = bounds.put(skip0.clone().requires_grad_(True), bounds.from_value(skip0.abs().max()))
        low = low0.clone().requires_grad_(True)
        lb = bounds.from_value(low0.abs().max())
        wa, wb_ = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
        import contextlib
        with (lazybn.scope() if lazy else contextlib.nullcontext()), fused.stat_groups(2):
            if fusedp:
                y, st = conv.conv2d_cat(skip, bounds.put(low, lb), wa, None, True, up2=True)
            else:
                up = fused.upsample2x(low)
                y, st = conv.conv2d_cat(skip, bounds.put(up, lb), wa, None, True)       # the same bound: the same power-of-two scale
            z = fused.bn_act_conv(y, st, None, bn, 0.01, wb_, None)
            z.backward(bounds.put(g.clone(), bounds.from_value(g.abs().max())))
        return [y.detach(), st.detach(), z.detach(), skip.grad, low.grad, wa.grad, wb_.grad]

    a, b = run(True), run(False)
    for i, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), (i, float((u - v).abs().max()))
    up_t = torch.nn.functional.interpolate(low0.cpu(), scale_factor=2, mode="bilinear", align_corners=True)
    torch.testing.assert_close(fused.upsample2x(low0).cpu(), up_t, rtol=1e-5, atol=2e-6)
    # without a bound on the low tensor the kernel refuses (UAPS_ENOFORM -> an error, not a silent fallback)
    with pytest.raises(Exception):
        conv.conv2d_cat(bounds.put(skip0.clone(), bounds.from_value(skip0.abs().max())), low0.clone(), w1, None, up2=True)


def test_batched_weight_gradient_reduction_equals_the_single_reductions():
    """uaps_conv_bwd_weight_reduce_batch over 31 gradients (two launches: 28 + 3) of mixed shapes -- 3x3 / 1x1, small / large
    element counts (both lane arrangements of the reduction), with / without a bias gradient -- equals
    uaps_conv_bwd_weight_reduce item by item, bit for bit."""
    import ctypes as C
    from uaps_amd import _lib, conv
    L = _lib.lib()
    DEV = torch.device("cuda:0")
    shapes = [(2, 16, 16, 32, 64, 3, True), (2, 64, 64, 16, 16, 3, False), (1, 128, 256, 16, 16, 3, True), (2, 256, 128, 16, 16, 1, False),
              (2, 3, 16, 24, 64, 3, True), (1, 20, 40, 18, 18, 3, False), (2, 16, 4, 32, 256, 3, True)]
    items, keep, singles = [], [], []
    st = _lib.current_stream(DEV)
    g = torch.Generator(device="cpu").manual_seed(3)
    for i in range(31):
        B, Cin, Cout, H, W, ks, bias = shapes[i % len(shapes)]
        x = torch.randn(B, Cin, H, W, generator=g).to(DEV)
        dy = torch.randn(B, Cout, H, W, generator=g).to(DEV)
        cfg = conv.plan_cfg(ks, 0, True, dy, x)
        n = C.c_size_t()
        _lib.check(L.uaps_conv_wrw_workspace_bytes(B, Cin, Cout, H, W, ks, cfg, C.byref(n)), "ws")
        ws = torch.empty(max(n.value, 16), dtype=torch.uint8, device=DEV)
        _lib.check(L.uaps_conv_bwd_weight_partial(dy.data_ptr(), x.data_ptr(), int(bias), B, Cin, Cout, H, W, ks, cfg, ws.data_ptr(), ws.numel(), st), "partial")
        dw1, db1 = torch.full((Cout, Cin, ks, ks), float("nan"), device=DEV), (torch.full((Cout,), float("nan"), device=DEV) if bias else None)
        _lib.check(L.uaps_conv_bwd_weight_reduce(ws.data_ptr(), dw1.data_ptr(), db1.data_ptr() if bias else None, B, Cin, Cout, H, W, ks, cfg, st), "reduce")
        dw2, db2 = torch.full_like(dw1, float("nan")), (torch.full_like(db1, float("nan")) if bias else None)
        items.append((ws, dw2, db2, B, Cin, Cout, H, W, ks, cfg))
        singles.append((dw1, db1))
        keep.append((x, dy))
    arr = (_lib.WrwReduceItem * len(items))()
    for a, (ws, dw, db, B, Cin, Cout, H, W, ks, cfg) in zip(arr, items):
        a.workspace, a.dw, a.dbias = ws.data_ptr(), dw.data_ptr(), (db.data_ptr() if db is not None else None)
        a.B, a.Cin, a.Cout, a.H, a.W, a.ks, a.cfg = B, Cin, Cout, H, W, ks, cfg
    _lib.check(L.uaps_conv_bwd_weight_reduce_batch(arr, len(items), st), "batch")
    torch.cuda.synchronize()
    for (dw1, db1), it in zip(singles, items):
        assert torch.isfinite(dw1).all()
        np.testing.assert_array_equal(it[1].cpu().numpy(), dw1.cpu().numpy())
        if db1 is not None:
            np.testing.assert_array_equal(it[2].cpu().numpy(), db1.cpu().numpy())
    assert L.uaps_conv_bwd_weight_reduce_batch(None, 0, st) == 0 and L.uaps_conv_bwd_weight_reduce_batch(None, 2, st) != 0


@pytest.mark.parametrize("B,H", [(2, 32), (3, 8)])
def test_whole_width_tile_form_equals_the_tile_kernels(B, H):
    """csrc/conv_split_g.hpp (round 6): 128-output-channel 3x3 layers on 32-wide maps as one workgroup per 4-row band and 128 channels
    (input staged once per layer, 32-channel chunks, weight fragments straight from L2, BatchNorm partials per 4-row band behind
    UAPS_CONV_BOUNDED) against the 8 x 32-tile kernels (UAPS_TUNE_NO_G): a two-source convolution with statistics -> BatchNorm ->
    LeakyReLU -> convolution with the normalisation in its staging -> BatchNorm, forward values and every gradient (the input
    gradients run 128 -> 128 and 128 -> 256 with two output tensors on the new form too).  Both are fp32-accurate; they differ by
    summation order only."""
    import torch.nn as nn
    from uaps_amd import _lib, bounds, conv, fused
    if conv.get_mode() != "h16":
        pytest.skip("the whole-width tile form exists in the fp16-split arithmetic only")
    dev = torch.device("cuda:0")
    torch.manual_seed(B * 10 + H)
    W = 32
    skip0 = torch.randn(B, 128, H, W, device=dev)
    up0 = torch.randn(B, 128, H, W, device=dev) * 1.5
    w1 = torch.randn(128, 256, 3, 3, device=dev) / 30
    w2 = torch.randn(128, 128, 3, 3, device=dev) / 20
    bn1, bn2 = nn.BatchNorm2d(128).to(dev), nn.BatchNorm2d(128).to(dev)
    with torch.no_grad():
        for bn in (bn1, bn2):
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
    g = torch.randn(B, 128, H, W, device=dev)
    L = _lib.lib()

    def run(no_g):
        L.uaps_conv_set_tuning(2048 if no_g else 0)
        conv._parts_cache.clear(); conv._variant_cache.clear()
        try:
            for bn in (bn1, bn2):
                bn.running_mean.zero_(); bn.running_var.fill_(1.0); bn.num_batches_tracked.zero_()
            bounds.refresh([bn1, bn2])
            a = bounds.put(skip0.clone().requires_grad_(True), bounds.from_value(skip0.abs().max()))
            b = bounds.put(up0.clone().requires_grad_(True), bounds.from_value(up0.abs().max()))
            p1, p2 = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
            y, st = conv.conv2d_cat(a, b, p1, None, with_stats=True)
            parts = int(st.shape[2])
            z, zst = fused.bn_act_conv(y, st, None, bn1, 0.01, p2, None, want_stats=True)
            out = fused.bn_act(z, None, bn2, 0.01, 0.0, True, stats=zst)
            (out * g).sum().backward()
            torch.cuda.synchronize()
            return parts, [t.detach().clone() for t in (y, z, out, a.grad, b.grad, p1.grad, p2.grad, bn1.weight.grad, bn1.bias.grad,
                                                          bn1.running_mean, bn1.running_var, bn2.running_var)]
        finally:
            L.uaps_conv_set_tuning(0)
            conv._parts_cache.clear(); conv._variant_cache.clear()
            for bn in (bn1, bn2):
                bn.weight.grad = bn.bias.grad = None

    parts_g, new = run(False)
    parts_t, old = run(True)
    assert parts_g == H // 4 and parts_t == (H + 7) // 8            # one part per 4-row band against one per 8 x 32 tile
    names = "y z out dskip dup dw1 dw2 dgamma dbeta rmean rvar rvar2".split()
    for n, u, v in zip(names, new, old):
        scale = float(v.abs().max()) + 1e-12
        assert float((u - v).abs().max()) <= 3e-5 * scale, (n, float((u - v).abs().max()) / scale)
