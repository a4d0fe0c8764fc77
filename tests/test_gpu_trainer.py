"""-m gpu: trainer-level behaviour on the GPU -- the supervised baseline step of BASELINE.json configs[0]
(baseline/baseline_train.py:158-173) against the CPU oracle, and the reference's per-batch metric averaging in the
training and validation loops (UAPS_train.py:305-306, 367-399)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_baseline_supervised_step_vs_cpu_oracle():
    """configs[0]: single-decoder U-Net, batch 4, 256 x 256, 4 classes; loss = 0.5 * (dice + CE), Adam.  One step through
    BaselineTrainer (HIP kernels) against the oracle's unfused restatement: logits, loss, every gradient, and the
    parameter delta of the Adam step."""
    import uaps_amd
    from oracle import uaps_oracle as O
    torch.manual_seed(21)
    rng = np.random.default_rng(21)
    B, H, W, C = 4, 256, 256, 4
    model = uaps_amd.UNet(3, C)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0                                        # the encoder's dropout draws are not injectable: off on both sides
    # float64 oracle: the weight gradients are sums of 10^5..10^6 products of mixed sign; an fp32 reference would carry as
    # much rounding error as the kernels under test
    sd = {k: (v.detach().clone().double() if v.is_floating_point() else v.detach().clone()) for k, v in model.state_dict().items()}
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k in sd:
        if k.endswith(".weight") or k.endswith(".bias"):
            sd[k].requires_grad_(True)
    x = torch.tensor(rng.standard_normal((B, 3, H, W)).astype(np.float32))
    y = torch.tensor(uaps_amd.data.synthetic_masks(rng, B, C, H, W))
    # oracle: Encoder.forward + Decoder.forward (UAPS_unet.py:110-116, 141-153), baseline_train.py:158-164
    feats = O.encoder_forward(x.double(), sd, "encoder", True, dropout=[0.0] * 5)
    logits_c = O.decoder_forward(feats, sd, "decoder", True)
    ce, dice = O.cross_entropy(logits_c, y), O.dice_loss(y.unsqueeze(1), logits_c)
    loss_c = 0.5 * (dice + ce)
    loss_c.backward()
    pkeys = [k for k in sd if sd[k].requires_grad]
    opt = torch.optim.Adam([sd[k] for k in pkeys], lr=1e-3)
    opt.step()

    model.to(DEV)
    tr = uaps_amd.BaselineTrainer(model, base_lr=1e-3)
    captured = {}
    hooks = [p.register_hook(lambda g, n=n: captured.__setitem__(n, g.detach().clone())) for n, p in model.named_parameters()]
    model.train()
    with torch.no_grad():
        pass
    res = tr.train_step(x.to(DEV), y.to(DEV))
    for h in hooks:
        h.remove()
    np.testing.assert_allclose(float(res["loss"]), float(loss_c), rtol=2e-5)
    np.testing.assert_allclose(float(res["ce"]), float(ce), rtol=2e-5)
    np.testing.assert_allclose(float(res["dice"]), float(dice), rtol=2e-5)
    lr, checked = 1e-3, 0
    for n, p in model.named_parameters():
        ref = sd[n].grad
        scale = float(ref.abs().max())
        err = float((captured[n].cpu().double() - ref).abs().max())
        if scale < 1e-7:
            assert err < 1e-6, n
        else:
            # 23 layers of fp32 sums in another order + LeakyReLU derivative flips at pre-activations within rounding of 0;
            # BatchNorm weight gradients are sums of ~10^5..10^6 terms of mixed sign: a single LeakyReLU derivative flip at a pre-activation within rounding of 0 moves an element by ~|dL/da * x| ~ 1e-6..1e-5: absolute floor 1e-5
            assert err <= max(5e-3 * scale, 1e-5), f"{n}: max err {err:.3e} vs scale {scale:.3e}"
        # Adam's first step moves an element by ~lr against the sign of its gradient
        sure = ref.abs() > max(1e-5, 1e-2 * scale)
        delta, delta_ref = p.detach().cpu() - init[n], (sd[n].detach() - init[n].double()).float()
        checked += int(sure.sum())
        if sure.any():
            assert float((delta[sure] - delta_ref[sure]).abs().max()) <= 0.02 * lr, n
    assert checked > 10000
    m = tr.epoch_metrics()
    ref_m = O.metrics_from_confusion(O.confusion(logits_c.detach().float(), y, C).numpy())
    for k in ("miou", "mdice", "acc"):
        assert abs(m[k] - ref_m[k]) < 1e-6 or (np.isnan(m[k]) and np.isnan(ref_m[k]))


def test_validation_and_training_metrics_average_per_batch_like_the_reference():
    """UAPSTrainer.validate / epoch_metrics: metrics of each batch averaged over the batches (UAPS_train.py:388-399), checked
    against the oracle's metric functions on the same eval-mode logits; a class missing from one batch makes the pooled
    number differ."""
    import uaps_amd
    from oracle import uaps_oracle as O
    torch.manual_seed(3)
    rng = np.random.default_rng(3)
    model = uaps_amd.net_factory("unet_uaps", 3, 4)
    tr = uaps_amd.UAPSTrainer(model)
    batches = []
    for i in range(3):
        x = torch.randn(2, 3, 64, 64, device=DEV)
        y = torch.tensor(rng.integers(0, 4, (2, 64, 64)), device=DEV)
        if i == 1:
            y[y == 3] = 0
        batches.append((x, y))
    m = tr.validate(batches)
    model.eval()
    per, ces = [], []
    with torch.no_grad():
        for x, y in batches:
            z = model(x)[0].cpu()
            per.append(O.metrics_from_confusion(O.confusion(z, y.cpu(), 4).numpy()))
            ces.append(float(O.cross_entropy(z, y.cpu())))
    for k in ("miou", "mdice", "acc"):
        assert abs(m[k] - np.mean([p[k] for p in per])) < 1e-9, k
    assert abs(m["ce"] - np.mean(ces)) < 1e-5
    assert abs(m["loss"] - np.mean([0.5 * ((1 - p["mdice"]) + c) for p, c in zip(per, ces)])) < 1e-5
    pooled = tr.validate(batches, pooled=True)
    assert abs(pooled["mdice"] - m["mdice"]) > 1e-9
    # training loop: one confusion matrix per step, averaged per batch at epoch end
    data = uaps_amd.data.SyntheticBatches(2, 3, 4, 64, 64, n_batches=2, device=DEV)
    for _ in range(3):
        tr.train_step(*data.next())
    assert len(tr._cms) == 3
    cms = torch.stack(tr._cms).cpu().numpy()
    em = tr.epoch_metrics()
    ref = uaps_amd.mean_batch_metrics(cms)
    assert all((np.isnan(em[k]) and np.isnan(ref[k])) or em[k] == ref[k] for k in em) and tr._cms == []


def test_fit_runs_the_epoch_loop_on_the_device_and_resumes(tmp_path):
    """UAPSTrainer.fit with the real device work (UAPS_train.py:279-450): 2 epochs of 3 steps on a small net, validation,
    best-checkpoint saving; then a second trainer resumes from the checkpoint and continues with the same iteration count,
    scheduler epoch and mixing-weight stream, and its first resumed step equals the uninterrupted run's step bit for bit
    when the model state is the checkpointed one."""
    import os
    import uaps_amd
    g = torch.Generator().manual_seed(5)

    def loader(n, seed_off):
        return [(torch.randn(2, 3, 64, 64, generator=g).to(DEV), torch.randint(0, 4, (2, 64, 64), generator=g).to(DEV)) for _ in range(n)]

    lab, unl, val = loader(2, 0), loader(3, 1), loader(2, 2)
    torch.manual_seed(3)
    model = uaps_amd.UNet_UAPS(3, 4, feature_chns=[16, 32, 32, 32, 32]).to(DEV)
    tr = uaps_amd.UAPSTrainer(model, base_lr=1e-3)
    path = os.path.join(tmp_path, "ck", "UAPS.pth")
    seen = []
    hist = tr.fit(lab, unl, val, epochs=3, iter_per_epoch=4, checkpoint_path=path, log=seen.append)
    assert [h["epoch"] for h in hist] == [1, 2] and tr.iter_num == 6 and seen == hist
    assert all(np.isfinite(h[k]) for h in hist for k in ("loss", "sup", "unsup", "val_mdice", "val_loss", "train_mdice"))
    assert hist[0]["saved"] and os.path.exists(path)                          # the first validation beats best_dice = 0
    ck = torch.load(path, weights_only=False)
    assert ck["epoch"] in (1, 2) and ck["iter_num"] == 3 * ck["epoch"] and ck["best_dice_1"] == max(h["val_mdice"] for h in hist[:ck["epoch"]])
    # resume
    model2 = uaps_amd.UNet_UAPS(3, 4, feature_chns=[16, 32, 32, 32, 32]).to(DEV)
    tr2 = uaps_amd.UAPSTrainer(model2, base_lr=1e-3)
    got = tr2.load_checkpoint(path)
    assert tr2.iter_num == ck["iter_num"] and tr2.scheduler.last_epoch == ck["epoch"]
    for (k1, v1), (k2, v2) in zip(ck["state_dict"].items(), model2.state_dict().items()):
        assert k1 == "module." + k2 and torch.equal(v1.to(DEV), v2)
    h2 = tr2.fit(lab, unl, val, epochs=ck["epoch"] + 2, iter_per_epoch=4, start_epoch=got["epoch"] + 1, best_dice=got["best_dice_1"])
    assert [h["epoch"] for h in h2] == [ck["epoch"] + 1] and tr2.iter_num == ck["iter_num"] + 3
    assert all(np.isfinite(h2[0][k]) for k in ("loss", "val_mdice"))
