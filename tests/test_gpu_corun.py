"""-m gpu: the packed-fp32 kernels beside 16x16x32 matrix kernels on a second stream.

On MI355X a `v_pk_fma_f32` / `v_pk_add_f32` whose second source takes its LOW half from the high register of a pair returns a
wrong low half for lanes 48..63 whenever another wave of the same SIMD is issuing `v_mfma_f32_16x16x32_f16`.  The shipped kernels
avoid that instruction form and tests/test_isa_lint.py keeps it out of the code objects by disassembly; this test is the guard
on the device itself: the kernels that do their arithmetic in packed fp32 (the exact-N class convolutions, forward with the
staging-time BatchNorm and weight gradient, and the loss block's forward) run a few thousand times on one stream while a second
stream keeps the chip busy with a 16x16x32 fp16 convolution, and every result must equal the first one bit for bit.  Before
the fix the out_conv launch deviated about once in a hundred launches under these conditions.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_packed_fp32_kernels_repeat_bit_for_bit_beside_16x16x32_matrix_kernels():
    import uaps_amd
    from uaps_amd import bounds, conv, fused, losses
    assert conv.get_mode() == "h16"
    torch.manual_seed(3)
    dev = torch.device(DEV)
    # stream B: 16 -> 16 channels on a 256 x 256 map in the fp16 two-piece form = conv_hp16_kernel<2>, v_mfma_f32_16x16x32_f16
    xb_ = torch.randn(8, 16, 256, 256, device=dev)
    wb_ = torch.randn(16, 16, 3, 3, device=dev) * 0.05
    wfb, _ = conv.pack_weights(wb_)
    bb = (bounds.from_value(xb_.abs().max()), 1.0)
    name = conv.kernel_variant("fwd", 8, 16, 16, 256, 256, 3)
    assert name.startswith("conv_sfwd_kernel<3, 8, 32, 16,"), name        # the plan the fp16 form turns into conv_hp16 / conv_hfwd (16x16x32)
    # stream A: out_conv of a decoder (16 -> 4 classes) with the previous BatchNorm applied while staging, its weight gradient,
    # and the loss forward on the logits
    B, Cc, H, W, Cout, G = 4, 16, 64, 64, 4, 2
    x = torch.randn(B, Cc, H, W, device=dev)
    w1 = torch.randn(Cc, Cc, 3, 3, device=dev) * 0.1
    wo = torch.randn(Cout, Cc, 3, 3, device=dev) * 0.1
    bn = torch.nn.BatchNorm2d(Cc).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.2, 0.2)
    bounds.refresh([bn])
    labels = torch.randint(0, Cout, (B // 2, H, W), device=dev)
    wmix = np.full(4, 0.25)

    def once():
        xx = x.clone().requires_grad_(True)
        woo = wo.clone().requires_grad_(True)
        bo = torch.zeros(Cout, device=dev, requires_grad=True)
        bn.running_mean.zero_(); bn.running_var.fill_(1.0)
        with fused.stat_groups(G):
            y, st = conv.conv2d_with_stats(xx, w1, None)
            z = fused.bn_act_conv(y, st, None, bn, 0.01, woo, bo)          # conv_small_bn_kernel<8, 4>
        heads = [z, z * 0.5, z * 0.25 + 0.1, -z]
        out = losses.uaps_pair_loss(tuple(heads), labels, wmix, 0.05, 0.07)    # pair_fwd_kernel / pair_bwd_kernel
        out.loss.backward()                                                  # ... conv_small_wrw_bn_kernel for woo.grad
        return [z.detach(), woo.grad.detach(), out.loss.detach().reshape(1), xx.grad.detach()]

    ref = [t.clone() for t in once()]
    side = torch.cuda.Stream(device=dev)
    bad = torch.zeros(len(ref), dtype=torch.int64, device=dev)
    n = 1500
    for i in range(n):
        with torch.cuda.stream(side):
            conv.conv_fwd_raw(xb_, wfb, None, 16, 3, 0, xb=bb)
        got = once()
        for k, (g, r) in enumerate(zip(got, ref)):
            bad[k] += (g.view(torch.int32) != r.view(torch.int32)).any()
    torch.cuda.synchronize()
    assert bad.tolist() == [0] * len(ref), f"launches (of {n}) whose [logits, out_conv weight gradient, loss, input gradient] deviated: {bad.tolist()}"


def test_column_strip_kernels_repeat_bit_for_bit_beside_other_launches():
    """The full-width-row kernels on 256-wide column strips (W = 512) write a strip's neighbour pixels into margin units that the
    kernel zeroes at its start.  The first build had no barrier between the zeroing and the first such store: a late zero wiped
    a neighbour pixel now and then -- never in single-stream runs here, a few times per hundred steps beside other streams'
    launches (seen as a final loss that differed from run to run at BASELINE.json configs[3]).  Forward with statistics, input
    gradient and weight gradient of a 512-wide layer, a few hundred times beside a second stream: every result equals the first."""
    from uaps_amd import bounds, conv
    assert conv.get_mode() == "h16"
    torch.manual_seed(5)
    dev = torch.device(DEV)
    xb_ = torch.randn(8, 64, 64, 64, device=dev)
    wb_ = torch.randn(64, 64, 3, 3, device=dev) * 0.05
    wfb, _ = conv.pack_weights(wb_)
    bb = (bounds.from_value(xb_.abs().max()), 1.0)
    B, Cin, Cout, H, W = 2, 16, 16, 64, 512
    x = torch.randn(B, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.1
    dy = torch.randn(B, Cout, H, W, device=dev)
    wf, wbk = conv.pack_weights(w)
    xbd, dyb = (bounds.from_value(x.abs().max()), 1.0), (bounds.from_value(dy.abs().max()), 1.0)
    conv.KERNEL_EVENTS, conv.EVENT_FILTER = {}, None
    conv.conv_fwd_raw(x, wf, None, Cout, 3, 0, want_stats=True, xb=xbd)
    conv.conv_bwd_weight_raw(dy, x, 3, False, 0, dyb=dyb, xb=xbd)
    names, conv.KERNEL_EVENTS = set(conv.KERNEL_EVENTS), None
    assert {"conv_hr16w_kernel<2>", "conv_hrwrww_kernel<1>"} <= names, names

    def once():
        y, st, _ = conv.conv_fwd_raw(x, wf, None, Cout, 3, 0, want_stats=True, xb=xbd)
        dx = conv.conv_bwd_data_raw(dy, wbk, Cin, 3, 0, dyb=dyb)
        dw, _ = conv.conv_bwd_weight_raw(dy, x, 3, False, 0, dyb=dyb, xb=xbd)
        return [y, st, dx, dw.clone()]

    ref = [t.clone() for t in once()]
    side = torch.cuda.Stream(device=dev)
    bad = torch.zeros(len(ref), dtype=torch.int64, device=dev)
    n = 400
    for i in range(n):
        with torch.cuda.stream(side):
            conv.conv_fwd_raw(xb_, wfb, None, 64, 3, 0, xb=bb)
        for k, (g, r) in enumerate(zip(once(), ref)):
            bad[k] += (g.view(torch.int32) != r.view(torch.int32)).any()
    torch.cuda.synchronize()
    assert bad.tolist() == [0] * len(ref), f"launches (of {n}) whose [output, statistics, input gradient, weight gradient] deviated: {bad.tolist()}"
