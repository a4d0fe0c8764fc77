"""-m gpu: the one-launch loss block (uaps_pairloss_*: both branches of UAPS_train.py:186-282 in one forward and one backward
launch) against the two-branch kernels it replaces, and its gathered-batch mode (the reference's nn.DataParallel computes
every mean / Dice sum over the logits of all GPUs, UAPS_model.py:13) against one launch on the concatenated batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("D,B,C,H,W", [(4, 16, 4, 256, 256), (6, 8, 2, 512, 512), (4, 2, 7, 64, 64), (3, 3, 5, 13, 17), (1, 2, 3, 9, 11), (8, 2, 8, 16, 16)])
def test_pair_kernels_equal_the_two_branch_kernels(D, B, C, H, W):
    """Pseudo-labels identical, variance maps to a few ulps, the reduced scalars and
    the gradients that depend on them to fp32 summation-order accuracy (the persistent pair kernels partition the pixels
    over blocks and threads differently), and the pair kernels themselves bitwise reproducible run to run."""
    import uaps_amd
    rng = np.random.default_rng(D * 100 + C)
    both = [torch.tensor((rng.standard_normal((2 * B, C, H, W)) * 2).astype(np.float32), device=DEV) for _ in range(D)]
    y = torch.tensor(rng.integers(0, C, (B, H, W)), device=DEV)
    w = rng.dirichlet(np.ones(D))
    a = [t.clone().requires_grad_(True) for t in both]
    o1 = uaps_amd.uaps_pair_loss(a, y, w, 0.06, 0.09, return_var=True)
    o1.loss.backward()
    lab = [t[:B].clone().requires_grad_(True) for t in both]
    un = [t[B:].clone().requires_grad_(True) for t in both]
    o2 = uaps_amd.uaps_step_loss(lab, y, un, w, 0.06, 0.09, return_var=True)
    o2.loss.backward()
    assert torch.equal(o1.pseudo, o2.pseudo)
    np.testing.assert_allclose(o1.var.cpu().numpy(), o2.var.cpu().numpy(), rtol=1e-5, atol=2e-6)     # instruction selection (fma contraction) differs between the instantiations
    np.testing.assert_allclose(float(o1.loss), float(o2.loss), rtol=2e-6)
    np.testing.assert_allclose(o1.sup_scalars.cpu().numpy(), o2.sup_scalars.cpu().numpy(), rtol=2e-5, atol=1e-7)
    u1, u2 = o1.unsup_scalars.cpu().numpy().copy(), o2.unsup_scalars.cpu().numpy()
    tot = 4 * D + 3                                  # UAPS_U_TOTAL: the pair finalize also writes supervised + consistency loss (round 6)
    assert u2[tot] == 0.0 and u1[tot] == np.float32(np.float32(o1.sup.cpu()) + np.float32(o1.unsup.cpu())) == np.float32(o1.loss.detach().cpu())
    u1[tot] = 0.0
    np.testing.assert_allclose(u1, u2, rtol=2e-5, atol=1e-7)
    for k in range(D):
        gmax = float(lab[k].grad.abs().max()) + float(un[k].grad.abs().max())
        np.testing.assert_allclose(a[k].grad[:B].cpu().numpy(), lab[k].grad.cpu().numpy(), rtol=1e-4, atol=1e-6 * gmax)
        np.testing.assert_allclose(a[k].grad[B:].cpu().numpy(), un[k].grad.cpu().numpy(), rtol=1e-4, atol=1e-6 * gmax)
    a2 = [t.clone().requires_grad_(True) for t in both]
    o3 = uaps_amd.uaps_pair_loss(a2, y, w, 0.06, 0.09, return_var=True)
    o3.loss.backward()
    assert torch.equal(o1.loss, o3.loss) and torch.equal(o1.sup_scalars, o3.sup_scalars) and torch.equal(o1.unsup_scalars, o3.unsup_scalars)
    assert all(torch.equal(a[k].grad, a2[k].grad) for k in range(D))


@pytest.mark.parametrize("D,B,C,H,W", [(4, 4, 4, 64, 64), (6, 2, 2, 128, 128)])
def test_gathered_batch_statistics_from_exchanged_sums(D, B, C, H, W):
    """Two 'ranks' simulated in one process: each runs the forward on its shard and contributes its raw sums, the sums are
    added (what the all-reduce does), both finalise with the global pixel count and run their backward.  Loss, scalars and
    logit gradients must equal ONE run on the concatenated batch -- the gathered-batch loss of the reference -- and differ
    from the per-shard loss (Dice is not additive over shards)."""
    import uaps_amd
    rng = np.random.default_rng(7)
    mk = lambda: [torch.tensor((rng.standard_normal((B, C, H, W)) * 2).astype(np.float32), device=DEV) for _ in range(D)]
    lab = [mk(), mk()]
    un = [mk(), mk()]
    y = [torch.tensor(rng.integers(0, C, (B, H, W)), device=DEV) for _ in range(2)]
    w = rng.dirichlet(np.ones(D))
    cw1, cw2 = 0.08, 0.03
    # reference: one process holds the gathered batch
    lab_all = [torch.cat([lab[0][k], lab[1][k]]).requires_grad_(True) for k in range(D)]
    un_all = [torch.cat([un[0][k], un[1][k]]).requires_grad_(True) for k in range(D)]
    ref = uaps_amd.uaps_step_loss(lab_all, torch.cat(y), un_all, w, cw1, cw2, exchange=lambda t: 1)
    ref.loss.backward()
    # pass 1 of both ranks: collect the raw sums (the exchange callable of rank r sees the OTHER rank's sums added)
    sums = []
    for r in range(2):
        def grab(t, r=r):
            sums.append(t.clone())
            return 1
        uaps_amd.uaps_step_loss([z.clone() for z in lab[r]], y[r], [z.clone() for z in un[r]], w, cw1, cw2, exchange=grab)
    total = sums[0] + sums[1]
    outs, grads = [], []
    for r in range(2):
        zl = [z.clone().requires_grad_(True) for z in lab[r]]
        zu = [z.clone().requires_grad_(True) for z in un[r]]

        def exch(t):
            t.copy_(total)
            return 2
        o = uaps_amd.uaps_step_loss(zl, y[r], zu, w, cw1, cw2, exchange=exch)
        o.loss.backward()
        outs.append(o)
        grads.append((zl, zu))
    for r in range(2):
        np.testing.assert_allclose(float(outs[r].loss), float(ref.loss), rtol=1e-6)
        np.testing.assert_allclose(outs[r].sup_scalars.cpu().numpy(), ref.sup_scalars.cpu().numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(outs[r].unsup_scalars.cpu().numpy(), ref.unsup_scalars.cpu().numpy(), rtol=1e-5, atol=1e-7)
        for k in range(D):
            gl, gu = grads[r][0][k].grad, grads[r][1][k].grad
            np.testing.assert_allclose(gl.cpu().numpy(), lab_all[k].grad[r * B:(r + 1) * B].cpu().numpy(), rtol=1e-4, atol=1e-10)
            np.testing.assert_allclose(gu.cpu().numpy(), un_all[k].grad[r * B:(r + 1) * B].cpu().numpy(), rtol=1e-4, atol=1e-10)
    local = uaps_amd.uaps_step_loss(lab[0], y[0], un[0], w, cw1, cw2)
    assert abs(float(local.loss) - float(ref.loss)) > 1e-5          # the per-shard loss is a different number


def test_gathered_loss_vs_cpu_oracle():
    """The exchanged-sums loss against the oracle's loss_from_sums on the CPU (values and gradients of one shard)."""
    import uaps_amd
    from oracle import uaps_oracle as O
    rng = np.random.default_rng(11)
    D, B, C, H, W = 4, 2, 4, 32, 32
    mk = lambda: [(rng.standard_normal((B, C, H, W)) * 2).astype(np.float32) for _ in range(D)]
    lab, un, lab2, un2 = mk(), mk(), mk(), mk()
    y, y2 = rng.integers(0, C, (B, H, W)), rng.integers(0, C, (B, H, W))
    w = rng.dirichlet(np.ones(D))
    cw1, cw2 = 0.1, 0.05
    t = lambda a: torch.tensor(a).double()
    labc, unc = [t(a).requires_grad_(True) for a in lab], [t(a).requires_grad_(True) for a in un]
    s1 = O.loss_sums(unc, labc, torch.tensor(y), w)
    s2 = O.loss_sums([t(a) for a in un2], [t(a) for a in lab2], torch.tensor(y2), w)
    tot = {k: s1[k] + s2[k].detach() for k in s1 if k != "pseudo"}
    ref = O.loss_from_sums(tot, 2 * B * H * W, cw1, cw2)
    ref["loss"].backward()
    g = lambda a, grad=False: torch.tensor(a, device=DEV).requires_grad_(grad)
    other = []
    uaps_amd.uaps_step_loss([g(a) for a in lab2], g(y2), [g(a) for a in un2], w, cw1, cw2, exchange=lambda s: (other.append(s.clone()), 1)[1])
    zl, zu = [g(a, True) for a in lab], [g(a, True) for a in un]
    o = uaps_amd.uaps_step_loss(zl, g(y), zu, w, cw1, cw2, exchange=lambda s: (s.add_(other[0]), 2)[1])
    o.loss.backward()
    np.testing.assert_allclose(float(o.loss), float(ref["loss"]), rtol=1e-5)
    for k in range(D):
        np.testing.assert_allclose(zl[k].grad.cpu().numpy(), labc[k].grad.numpy(), rtol=2e-3, atol=1e-9)
        np.testing.assert_allclose(zu[k].grad.cpu().numpy(), unc[k].grad.numpy(), rtol=2e-3, atol=1e-9)


def test_gradients_land_in_the_flat_bucket_buffers():
    """GradBuckets registers one flat buffer per bucket as the destination of its parameters' gradients: after a backward
    through the product path every .grad must already BE its slice of the flat buffer (no concatenation, no copy)."""
    import uaps_amd
    from uaps_amd import dist as udist
    torch.manual_seed(4)
    model = uaps_amd.net_factory("unet_uaps", 3, 4)
    buckets = udist.GradBuckets(model)
    try:
        assert buckets.names == ["aux_decoder3", "aux_decoder2", "aux_decoder1", "main_decoder", "encoder"]
        data = uaps_amd.data.SyntheticBatches(2, 3, 4, 64, 64, n_batches=1, device=DEV)
        xl, yl, xu = data.next()
        both = model.forward_pair(xl, xu)
        uaps_amd.uaps_pair_loss(both, yl, np.full(4, 0.25), 0.1, 0.1).loss.backward()
        n = 0
        for bi, params in enumerate(buckets.buckets):
            for k, p in enumerate(params):
                assert p.grad is not None and p.grad.data_ptr() == buckets._view(bi, k).data_ptr(), (buckets.names[bi], k)
                assert p.grad.data_ptr() % 16 == 0
                n += 1
        assert n == 208
    finally:
        buckets.remove()
