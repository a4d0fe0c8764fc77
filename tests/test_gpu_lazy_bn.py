"""-m gpu: the BatchNorm backward in two halves (uaps_amd/lazybn.py; include/uaps_hip.h: uaps_bn_act_bwd_prepare / _apply,
uaps_call_hints::dyt_*): the reductions in the node that owns the BatchNorm, dy formed by the weight-gradient kernel of the
convolution in front while it stages that operand (UAPS_unet.py:36-44 under autograd)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bn_state(C, groups, B, H, W, seed):
    from uaps_amd import fused
    g = torch.Generator().manual_seed(seed)
    y = (torch.randn(B, C, H, W, generator=g) * 2 + 0.3).to(DEV)
    dout = torch.randn(B, C, H, W, generator=g).to(DEV)
    bn = nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
    yr = y.clone().requires_grad_(True)
    with fused.stat_groups(groups):
        out = fused.bn_act(yr, None, bn, 0.01, 0.0, True)
    out.backward(dout)
    return y, dout, bn, yr.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()


@pytest.mark.parametrize("B,C,H,W,groups", [(4, 16, 32, 256, 2), (2, 24, 20, 36, 1), (6, 8, 7, 9, 3)])
def test_prepare_and_apply_equal_the_one_piece_backward(B, C, H, W, groups):
    """uaps_bn_act_bwd_prepare + uaps_bn_act_bwd_apply against uaps_bn_act_bwd_grouped (through fused.bn_act): dy, dgamma, dbeta bit
    for bit (the same arithmetic in the same order), and the bound `prepare` raises is an upper bound of max|dy|."""
    from uaps_amd import _lib, bounds, fused, lazybn
    y, dout, bn, dy_ref, dg_ref, db_ref = _bn_state(C, groups, B, H, W, 3)
    Bg = B // groups
    mean = torch.stack([y[g * Bg:(g + 1) * Bg].double().mean((0, 2, 3)) for g in range(groups)]).float()
    # the saved statistics of the forward: recompute through the library for bit equality
    from uaps_amd.fused import _bn_ws
    bn2 = nn.BatchNorm2d(C).to(DEV)
    bn2.load_state_dict(bn.state_dict())
    stats = torch.empty((2, groups * C), dtype=torch.float32, device=DEV)
    out = torch.empty_like(y)
    ws = _bn_ws(torch.device(DEV), B, C, H, W)
    L = _lib.lib()
    with _lib.device_guard(torch.device(DEV)):
        rc = L.uaps_bn_act_fwd_train_grouped(y.data_ptr(), None, bn2.weight.data_ptr(), bn2.bias.data_ptr(), None, None, None, 0.1, bn2.eps,
                                             0.01, 0.0, 0, 0, B, C, H, W, groups, out.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                                             ws.data_ptr(), ws.numel(), _lib.current_stream(torch.device(DEV)))
    _lib.check(rc, "fwd")
    dgamma, dbeta, dcb = (torch.empty(C, device=DEV) for _ in range(3))
    lz = lazybn.prepare(dout, y, bn2.weight, bn2.bias, stats[0], stats[1], 0.01, groups, dgamma, dbeta, dcb, ws)
    assert lazybn.take(dout) is lz and lazybn.take(dout) is None
    dy = lazybn.materialize(dout, lz)
    assert torch.equal(dy, dy_ref) and torch.equal(dgamma, dg_ref) and torch.equal(dbeta, db_ref)
    assert float(dcb.abs().max()) == 0.0
    upper = float(bounds.value(lz.bound[0])) * lz.bound[1]
    assert float(dy.abs().max()) <= upper <= 64.0 * float(dy.abs().max())
    assert abs(float(mean.abs().max())) >= 0.0          # (the statistics groups were exercised)


def _block(seed, C0, C1, H, W, B, lazy, cat, scope=True, hook=None, second_consumer=False, fail_in_backward=False):
    """conv1 (one or two tensors) -> BatchNorm -> LeakyReLU -> conv2 -> BatchNorm -> LeakyReLU -> 3x3 to 4 classes, the decoder's chain of
    nodes (ConvBlock + out_conv); returns the loss-free gradients of everything.  scope: forward + backward inside lazybn.scope()
    (what the trainers do); hook: called on the gradient of conv1's raw output; second_consumer: that output also feeds a plain
    sum; fail_in_backward: conv2's input gradient raises."""
    import contextlib
    from uaps_amd import bounds, conv, fused, lazybn
    lazybn._ON = lazy
    torch.manual_seed(seed)
    xs = [torch.randn(B, C0, H, W, device=DEV, requires_grad=True) for _ in range(2 if cat else 1)]
    w1 = (torch.randn(C1, C0 * len(xs), 3, 3, device=DEV) / 10).requires_grad_(True)
    w2 = (torch.randn(C1, C1, 3, 3, device=DEV) / 10).requires_grad_(True)
    w3 = (torch.randn(4, C1, 3, 3, device=DEV) / 10).requires_grad_(True)
    bn1, bn2 = nn.BatchNorm2d(C1).to(DEV), nn.BatchNorm2d(C1).to(DEV)
    with torch.no_grad():
        for bn in (bn1, bn2):
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
    bounds.refresh([bn1, bn2])
    xin = [bounds.put(x, bounds.from_value(x.detach().abs().max())) for x in xs]
    real_bwd_data, calls = conv.conv_bwd_data_raw, [0]

    def failing(*a, **k):
        calls[0] += 1
        if calls[0] == 2:
            raise RuntimeError("injected failure in the middle of a backward")
        return real_bwd_data(*a, **k)

    try:
        with (lazybn.scope() if scope else contextlib.nullcontext()):
            with fused.stat_groups(2):
                if cat:
                    y1, st1 = conv.conv2d_cat(xin[0], xin[1], w1, None, with_stats=True)
                else:
                    y1, st1 = conv.conv2d_with_stats(xin[0], w1, None)
                if hook is not None:
                    y1.register_hook(hook)
                y2, st2 = fused.bn_act_conv(y1, st1, None, bn1, 0.01, w2, None, want_stats=True)
                z = fused.bn_act_conv(y2, st2, None, bn2, 0.01, w3, None)
            g = torch.randn(z.shape, generator=torch.Generator().manual_seed(seed + 1)).to(DEV)
            if fail_in_backward:
                conv.conv_bwd_data_raw = failing
            if second_consumer:
                torch.autograd.backward([z, (y1 * 0.5).sum()], [bounds.put(g, bounds.from_value(g.abs().max())), None])
            else:
                z.backward(bounds.put(g, bounds.from_value(g.abs().max())))
    finally:
        conv.conv_bwd_data_raw = real_bwd_data
        lazybn._ON = True
    return [x.grad for x in xs] + [w1.grad, w2.grad, w3.grad, bn1.weight.grad, bn1.bias.grad, bn2.weight.grad, bn2.bias.grad]


@pytest.mark.parametrize("C0,C1,H,W,B,cat", [(16, 16, 32, 256, 4, True), (16, 16, 16, 256, 2, False), (32, 32, 32, 64, 4, True),
                                             (16, 32, 16, 16, 2, False), (16, 16, 16, 512, 2, True)])
def test_chain_gradients_with_and_without_the_pending_transform(C0, C1, H, W, B, cat):
    """The same chain of nodes with UAPS_LAZY_BN_BWD on and off: every gradient agrees to the rounding of the fp16-split weight
    gradient (dy formed in the kernel's staging is the same fp32 values; its operand bound is an upper bound instead of the exact
    maximum, so the power-of-two scale may differ)."""
    a = _block(7, C0, C1, H, W, B, True, cat)
    b = _block(7, C0, C1, H, W, B, False, cat)
    for u, v in zip(a, b):
        scale = float(v.abs().max()) + 1e-12
        assert float((u - v).abs().max()) <= 2e-5 * scale, float((u - v).abs().max()) / scale


def test_a_pending_transform_that_nobody_applies_is_an_error():
    from uaps_amd import lazybn
    lazybn._loose.outstanding = 1
    with pytest.raises(RuntimeError, match="pending BatchNorm transform"):
        lazybn.assert_none_pending()
    lazybn.assert_none_pending()


def test_the_two_halves_backward_runs_only_inside_a_scope():
    """A user-driven forward / loss.backward() (UAPS_train.py:177-292 on this package) never sees an
    untransformed gradient: outside lazybn.scope() nothing is handed up, and the gradients are the one-piece ones bit for bit."""
    from uaps_amd import lazybn
    n0 = lazybn.prepared_total()
    a = _block(11, 16, 16, 32, 256, 4, True, True, scope=False)
    assert lazybn.prepared_total() == n0
    b = _block(11, 16, 16, 32, 256, 4, False, True, scope=False)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    _block(11, 16, 16, 32, 256, 4, True, True, scope=True)
    assert lazybn.prepared_total() > n0              # (and inside a scope it does run at this shape)


def test_a_hook_on_a_marked_conv_output_sees_the_true_gradient():
    """register_hook on the raw output of a convolution whose gradient would be handed up untransformed: the observer gets dy itself
    (the node falls back to the one-piece backward for that tensor), bit for bit what it gets with the mechanism off."""
    from uaps_amd import lazybn
    seen = {}
    a = _block(5, 16, 16, 32, 256, 4, True, False, hook=lambda g: seen.__setitem__("lazy", g.detach().clone()))
    b = _block(5, 16, 16, 32, 256, 4, False, False, hook=lambda g: seen.__setitem__("plain", g.detach().clone()))
    assert torch.equal(seen["lazy"], seen["plain"])
    for u, v in zip(a, b):
        scale = float(v.abs().max()) + 1e-12
        assert float((u - v).abs().max()) <= 2e-5 * scale
    lazybn.assert_none_pending()


def test_a_second_consumer_of_a_marked_output_fails_the_step_instead_of_training_on_a_wrong_gradient():
    """The raw conv output also feeds a plain sum: autograd adds that gradient to the untransformed d(activation).  In place (same
    tensor, higher version) the producer refuses it; out of place the record is lost and the scope's exit check fails the step."""
    from uaps_amd import lazybn
    with pytest.raises(RuntimeError, match="pending BatchNorm transform"):
        _block(9, 16, 16, 32, 256, 4, True, False, second_consumer=True)
    lazybn.reset()
    lazybn.assert_none_pending()


def test_a_failed_backward_leaves_nothing_behind_for_a_plain_user_backward():
    """A backward that throws midway inside a trainer step, then the reference's own loop on the
    same process -- forward, loss.backward() outside any scope: the one-piece gradients bit for bit."""
    from uaps_amd import lazybn
    ref = _block(13, 16, 16, 32, 256, 4, False, True, scope=False)
    with pytest.raises(RuntimeError, match="injected failure"):
        _block(13, 16, 16, 32, 256, 4, True, True, scope=True, fail_in_backward=True)
    assert lazybn.current() is None
    got = _block(13, 16, 16, 32, 256, 4, True, True, scope=False)
    for u, v in zip(got, ref):
        assert torch.equal(u, v)
    lazybn.assert_none_pending()


@pytest.mark.parametrize("B,Cin,C,H,W", [(4, 16, 16, 32, 256), (4, 32, 16, 32, 256), (8, 32, 32, 128, 128), (2, 16, 16, 32, 512), (2, 32, 16, 16, 768)])
def test_dy_formed_in_the_weight_gradient_kernel_is_the_stand_alone_dy_bit_for_bit_and_repeats(B, Cin, C, H, W):
    """The kernel that forms dy while staging must write exactly what uaps_bn_act_bwd_apply writes, every time (a tile form of it,
    built and dropped in round 4, first compiled its transform into packed fp32 instructions of the operand form that misbehaves
    beside 16x16x32 matrix instructions, DESIGN.md section 4: a few hundred wrong elements per launch, different ones each run --
    tools/isa_lint.py flagged it); a layer without such a kernel takes the stand-alone pass."""
    from uaps_amd import _lib, bounds, conv, lazybn
    from uaps_amd.fused import _bn_ws
    dev = torch.device(DEV)
    torch.manual_seed(1)
    groups = 2
    x = torch.randn(B, Cin, H, W, device=dev)
    y = torch.randn(B, C, H, W, device=dev) * 2 + 0.3
    dout = torch.randn(B, C, H, W, device=dev)
    bn = nn.BatchNorm2d(C).to(dev)
    stats = torch.empty((2, groups * C), device=dev)
    out = torch.empty_like(y)
    ws = _bn_ws(dev, B, C, H, W)
    L = _lib.lib()
    with _lib.device_guard(dev):
        rc = L.uaps_bn_act_fwd_train_grouped(y.data_ptr(), None, bn.weight.data_ptr(), bn.bias.data_ptr(), None, None, None, 0.1, bn.eps, 0.01, 0.0,
                                             0, 0, B, C, H, W, groups, out.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), ws.data_ptr(),
                                             ws.numel(), _lib.current_stream(dev))
    _lib.check(rc, "fwd")
    dg, db, dc = (torch.empty(C, device=dev) for _ in range(3))
    xb = (bounds.from_value(x.abs().max()), 1.0)
    first = None
    for _ in range(3):
        lz = lazybn.prepare(dout, y, bn.weight, bn.bias, stats[0], stats[1], 0.01, groups, dg, db, dc, ws)
        assert lazybn.take(dout) is lz
        ref = lazybn.materialize(dout, lz)
        conv.KERNEL_EVENTS, conv.EVENT_FILTER = {}, None
        dw, _, dyt = conv.conv_bwd_weight_raw(dout, x, 3, False, 0, xb=xb, lz=lz)
        names, conv.KERNEL_EVENTS = set(conv.KERNEL_EVENTS), None
        assert any("_dt_kernel" in n for n in names) == (W % 256 == 0), names      # the in-kernel form on 256-wide maps and strips, elsewhere the fall-back
        assert torch.equal(dyt, ref)
        if first is None:
            first = dw.clone()
        assert torch.equal(dw, first)


def test_backward_sums_in_the_input_gradient_epilogue_match_the_sums_pass():
    """uaps_call_hints::bsum_* (round 5): the 16 -> 16 input gradient on a 256-wide map forms the backward sums of the BatchNorm in
    front of the convolution in its epilogue (conv_hr16_bs_kernel) + uaps_bn_act_bwd_finalize, against uaps_bn_act_bwd_prepare's own
    pass over (gradient, y): the same input gradient bit for bit, coefficients / dgamma / dbeta to the rounding of differently
    grouped fp32 partial sums, a dy bound that is an upper bound; two launches give the same bits."""
    from uaps_amd import _lib, bounds, conv, fused, lazybn
    if conv.get_mode() != "h16":
        pytest.skip("fp16-split arithmetic only")
    dev = torch.device(DEV)
    torch.manual_seed(21)
    B, Cc, H, W, groups = 4, 16, 32, 256, 2
    y = (torch.randn(B, Cc, H, W, device=dev) * 1.5 + 0.2)
    dz = torch.randn(B, Cc, H, W, device=dev)
    w = torch.randn(Cc, Cc, 3, 3, device=dev) / 10
    _, wb = conv.pack_weights(w)
    bn = nn.BatchNorm2d(Cc).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.4, 0.4)
    Bg = B // groups
    mean = torch.stack([y[g * Bg:(g + 1) * Bg].mean((0, 2, 3)) for g in range(groups)]).contiguous()
    var = torch.stack([y[g * Bg:(g + 1) * Bg].var((0, 2, 3), unbiased=False) for g in range(groups)])
    invstd = (var + bn.eps).rsqrt().contiguous()
    dzb = (bounds.from_value(dz.abs().max()), 1.0)
    plain = conv.conv_bwd_data_raw(dz, wb, Cc, 3, 0, dyb=dzb)
    prev, conv._FUSED_BSUM = conv._FUSED_BSUM, True       # (not the default: profiles/r05_bn_sums_epilogue_ab.txt)
    da, partials, maxes = conv.conv_bwd_data_raw(dz, wb, Cc, 3, 0, dyb=dzb, bsum=(y, mean, invstd, bn.weight, bn.bias, 0.01, groups))
    assert partials is not None and torch.equal(da, plain)
    da2, partials2, maxes2 = conv.conv_bwd_data_raw(dz, wb, Cc, 3, 0, dyb=dzb, bsum=(y, mean, invstd, bn.weight, bn.bias, 0.01, groups))
    conv._FUSED_BSUM = prev
    assert torch.equal(partials, partials2) and torch.equal(maxes, maxes2)
    dg1, db1, dc1 = (torch.empty(Cc, device=dev) for _ in range(3))
    dg2, db2, dc2 = (torch.empty(Cc, device=dev) for _ in range(3))
    ws = fused._bn_ws(dev, B, Cc, H, W)
    lz_ref = lazybn.prepare(da, y, bn.weight, bn.bias, mean, invstd, 0.01, groups, dg1, db1, dc1, ws)
    assert lazybn.take(da) is lz_ref
    lz = lazybn.prepare_from_partials(da, y, bn.weight, bn.bias, mean, invstd, 0.01, groups, dg2, db2, dc2, partials, maxes)
    assert lazybn.take(da) is lz
    torch.testing.assert_close(lz.coef, lz_ref.coef, rtol=2e-5, atol=1e-7)
    torch.testing.assert_close(dg2, dg1, rtol=2e-5, atol=1e-3)
    torch.testing.assert_close(db2, db1, rtol=2e-5, atol=1e-3)
    b_ref, b_new = float(bounds.value(lz_ref.bound[0])), float(bounds.value(lz.bound[0]))
    dy = lazybn.materialize(da, lz)
    assert float(dy.abs().max()) <= b_new and abs(b_new - b_ref) <= 1e-4 * b_ref
    lazybn.reset()
