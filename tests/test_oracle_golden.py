"""Pins the CPU oracle (oracle/) against the fixtures generated from the imported reference."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import uaps_oracle as O

from conftest import GOLDEN

G1 = sorted(glob.glob(os.path.join(GOLDEN, "g1_*.npz")))
assert G1, "golden fixtures missing"


def _load(p):
    return np.load(p, allow_pickle=False)


@pytest.mark.parametrize("path", G1, ids=[os.path.basename(p)[3:-4] for p in G1])
def test_torch_oracle_loss_block(path):
    g = _load(path)
    D = int(g["D"])
    w = g["w"]
    for tag in ("r0", "full", "mt"):
        cw1, cw2 = g[f"cw_{tag}"]
        un = [torch.tensor(g["un_logits"][k], requires_grad=True) for k in range(D)]
        lab = [torch.tensor(g["lab_logits"][k], requires_grad=True) for k in range(D)]
        r = O.step_loss(un, lab, torch.tensor(g["labels"]), w, float(cw1), float(cw2))
        r["loss"].backward()
        np.testing.assert_allclose(r["loss"].item(), g[f"loss_{tag}"], rtol=2e-6, atol=1e-6)
        gu = np.stack([t.grad.numpy() for t in un])
        gl = np.stack([t.grad.numpy() for t in lab])
        ref_u = g[f"g_un_{tag}"]
        fin = np.isfinite(ref_u)
        # the oracle mirrors the reference's torch ops, so even its NaNs (m_c == 0) agree
        assert np.array_equal(np.isfinite(gu), fin)
        np.testing.assert_allclose(gu[fin], ref_u[fin], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(gl, g[f"g_lab_{tag}"], rtol=1e-5, atol=1e-9)
        if tag == "full":
            assert np.array_equal(r["pseudo"].numpy(), g["pseudo"])
            np.testing.assert_allclose(np.stack([t.detach().numpy() for t in r["p"]]), g["un_soft"], rtol=0, atol=3e-7)
            np.testing.assert_allclose(r["m"].detach().numpy(), g["preds"], rtol=0, atol=3e-7)
            np.testing.assert_allclose(np.stack([t.detach().numpy() for t in r["var"]]), g["var"], rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(np.stack([t.detach().numpy() for t in r["evar"]]), g["evar"], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(r["mixed"].numpy(), g["mixed"], rtol=0, atol=3e-7)
            for key, gk in (("ce", "ce_ps"), ("dice", "dice_ps"), ("s", "ps"), ("psl", "psl"), ("ce_sup", "ce_sup"), ("dice_sup", "dice_sup")):
                np.testing.assert_allclose([float(t.detach()) for t in r[key]], g[gk], rtol=2e-6, atol=1e-7)
            np.testing.assert_allclose(float(r["l_uncert"]), g["l_uncert"], rtol=2e-6, atol=1e-7)
            np.testing.assert_allclose(float(r["ps_loss"]), g["ps_loss"], rtol=2e-6)
            np.testing.assert_allclose(float(r["sup"]), g["sup"], rtol=2e-6)


@pytest.mark.parametrize("path", G1, ids=[os.path.basename(p)[3:-4] for p in G1])
def test_c_oracle_loss_block(path):
    """The plain-C double-precision restatement incl. the closed-form gradient vs reference autograd."""
    g = _load(path)
    D, B, C, H, W = (int(g[k]) for k in "DBCHW")
    N = B * H * W
    un = [g["un_logits"][k] for k in range(D)]
    lab = [g["lab_logits"][k] for k in range(D)]
    f = c_oracle.unsup_fwd(un, g["w"])
    margin = np.sort(g["mixed"], axis=1)
    margin = margin[:, -1] - margin[:, -2]
    clear = margin > 1e-5
    assert np.array_equal(f["pseudo"][clear], g["pseudo"][clear])
    # the rest of the comparison follows the reference's labels (ties may legitimately differ)
    if not np.array_equal(f["pseudo"], g["pseudo"]):
        assert "neartie" in path or "saturated" in path
        pytest.skip("tie-broken labels differ from the fp32 reference on exact-tie pixels; covered by torch oracle")
    np.testing.assert_allclose(f["mixed"], g["mixed"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(f["var"], g["var"], rtol=2e-5, atol=2e-6)
    lo = c_oracle.unsup_losses(f["stats"], D, C, N)
    np.testing.assert_allclose(lo["ce"], g["ce_ps"], rtol=1e-5)
    np.testing.assert_allclose(lo["dice"], g["dice_ps"], rtol=1e-5)
    np.testing.assert_allclose(lo["ps_loss"], g["ps_loss"], rtol=1e-5)
    np.testing.assert_allclose(lo["l_uncert"], g["l_uncert"], rtol=1e-5, atol=1e-7)
    ss = c_oracle.sup_fwd(lab, g["labels"])
    sl = c_oracle.sup_losses(ss, D, C, N)
    np.testing.assert_allclose(sl["sup"], g["sup"], rtol=1e-5)
    for tag in ("r0", "full", "mt"):
        cw1, cw2 = g[f"cw_{tag}"]
        gu = np.stack(c_oracle.unsup_bwd(un, f["pseudo"], f["stats"], cw1, cw2))
        ref = g[f"g_un_{tag}"]
        fin = np.isfinite(ref)
        assert np.isfinite(gu).all()           # documented deviation: finite where the reference NaNs
        scale = np.abs(ref[fin]).max()
        np.testing.assert_allclose(gu[fin], ref[fin], rtol=2e-4, atol=2e-6 * scale)
        gl = np.stack(c_oracle.sup_bwd(lab, g["labels"], ss))
        refl = g[f"g_lab_{tag}"]
        np.testing.assert_allclose(gl, refl, rtol=2e-4, atol=2e-6 * np.abs(refl).max())
        total = sl["sup"] + cw1 * lo["ps_loss"] + cw2 * lo["l_uncert"]
        np.testing.assert_allclose(total, g[f"loss_{tag}"], rtol=1e-5)


def test_ramp_and_scalar_losses():
    g = _load(os.path.join(GOLDEN, "g2_losses.npz"))
    for i, t in enumerate(g["ramp_t"]):
        for j, R in enumerate(g["ramp_R"]):
            assert abs(O.sigmoid_rampup(t, R) - g["ramp"][i, j]) < 1e-12
    for C in (4, 7, 2):
        a, b, y = (torch.tensor(g[f"{k}{C}"]) for k in "aby")
        np.testing.assert_allclose(float(O.dice_loss(y.unsqueeze(1), a)), g[f"dice{C}"], rtol=1e-6)
        np.testing.assert_allclose(float(O.dice_loss(y.unsqueeze(1), a, eps=1e-3)), g[f"dice{C}_eps"], rtol=1e-6)
        np.testing.assert_allclose(float(O.cross_entropy(a, y)), g[f"ce{C}"], rtol=1e-6)
        np.testing.assert_allclose(float(O.softmax_kl_mean(a, b)), g[f"softmax_kl{C}"], rtol=1e-5)
        np.testing.assert_allclose(O.softmax_mse_map(a, b).numpy(), g[f"softmax_mse{C}"], atol=1e-7)
        p, q = torch.softmax(a, 1), torch.softmax(b, 1)
        np.testing.assert_allclose(O.entropy_map(p).numpy(), g[f"entropy_map{C}"], atol=1e-6)
        np.testing.assert_allclose(float(O.kl_loss_probs(p, q)), g[f"kl_loss{C}"], rtol=1e-5)


def test_perturbations():
    g = _load(os.path.join(GOLDEN, "g3_perturb.npz"))
    for i in range(3):
        x = torch.tensor(g[f"x{i}"])
        assert np.array_equal(O.feature_noise(x, torch.tensor(g[f"noise{i}"])).numpy(), g[f"noise_y{i}"])
        assert np.array_equal(O.feature_dropout(x, float(g[f"fd_u{i}"])).numpy(), g[f"fd_y{i}"])
        assert np.array_equal(O.feature_bernoulli(x, torch.tensor(g[f"bern_mask{i}"])).numpy(), g[f"bern_y{i}"])


def test_metrics():
    g = _load(os.path.join(GOLDEN, "g5_metrics.npz"))
    for i in range(4):
        lg, y = g[f"logits{i}"], g[f"labels{i}"]
        cm = O.confusion(torch.tensor(lg), torch.tensor(y), 4).numpy()
        assert np.array_equal(cm, c_oracle.confusion(lg, y))
        m = O.metrics_from_confusion(cm)
        for key, gk in (("miou", "miou"), ("mdice", "mdice"), ("acc", "acc")):
            ref = float(g[f"{gk}{i}"])
            if np.isnan(ref):
                assert np.isnan(m[key])
            else:
                assert abs(m[key] - ref) < 1e-12


def test_functional_model_blocks_and_narrow_net():
    g = _load(os.path.join(GOLDEN, "g4_model.npz"))
    def sd(prefix):
        return {k[len(prefix):]: torch.tensor(g[k]) for k in g.files if k.startswith(prefix)}
    s = sd("cb."); x = torch.tensor(g["cb_x"])
    s2 = {"blk." + k: v.clone() for k, v in s.items()}
    np.testing.assert_allclose(O.conv_block(x, s2, "blk", True).numpy(), g["cb_y_train"], atol=2e-6)
    after = sd("cb_after.")
    for k in ("conv_conv.1.running_mean", "conv_conv.1.running_var", "conv_conv.5.running_mean", "conv_conv.5.running_var"):
        np.testing.assert_allclose(s2["blk." + k].numpy(), after[k].numpy(), atol=1e-6)
    np.testing.assert_allclose(O.conv_block(x, s2, "blk", False).numpy(), g["cb_y_eval"], atol=2e-6)
    # UpBlock
    s = {"up." + k: v for k, v in sd("ub.").items()}
    y = O.up_block(torch.tensor(g["ub_x1"]), torch.tensor(g["ub_x2"]), s, "up", True)   # also updates running stats
    np.testing.assert_allclose(y.numpy(), g["ub_y_train"], atol=2e-6)
    y = O.up_block(torch.tensor(g["ub_x1"]), torch.tensor(g["ub_x2"]), s, "up", False)
    np.testing.assert_allclose(y.numpy(), g["ub_y_eval"], atol=2e-6)
    # DownBlock = max-pool + ConvBlock
    s = {"dn.maxpool_conv.1." + k[len("maxpool_conv.1."):]: v for k, v in sd("db.").items()}
    import torch.nn.functional as F
    xd = F.max_pool2d(torch.tensor(g["db_x"]), 2)
    np.testing.assert_allclose(O.conv_block(xd, s, "dn.maxpool_conv.1", True).numpy(), g["db_y_train"], atol=2e-6)
    np.testing.assert_allclose(O.conv_block(xd, s, "dn.maxpool_conv.1", False).numpy(), g["db_y_eval"], atol=2e-6)
    # narrow encoder + main decoder, eval
    s = sd("narrow.")
    f = O.encoder_forward(torch.tensor(g["narrow_x"]), s, "encoder", False)
    for i in range(5):
        np.testing.assert_allclose(f[i].numpy(), g[f"narrow_feat{i}"], atol=5e-6)
    y = O.decoder_forward(f, s, "main_decoder", False)
    np.testing.assert_allclose(y.numpy(), g["narrow_y_eval"], atol=1e-5)


def test_full_step_g6():
    """One whole step (two forwards, loss, backward, Adam) of the functional oracle vs the reference classes."""
    g = _load(os.path.join(GOLDEN, "g6_step.npz"))
    init = {k[5:]: torch.tensor(g[k]) for k in g.files if k.startswith("init.")}
    st = O.CpuStep(init, lr=1e-3)
    rl = {"noise": [torch.tensor(g[f"noise_l{i}"]) for i in range(5)], "mask": [torch.tensor(g[f"mask_l{i}"]).float() for i in range(5)],
          "u": [float(g[f"u_l{i}"]) for i in range(5)]}
    ru = {"noise": [torch.tensor(g[f"noise_u{i}"]) for i in range(5)], "mask": [torch.tensor(g[f"mask_u{i}"]).float() for i in range(5)],
          "u": [float(g[f"u_u{i}"]) for i in range(5)]}
    zero = (0.0,) * 5
    lab = O.uaps_forward(torch.tensor(g["xl"]), st.sd, True, rl, dropout=zero)
    un = O.uaps_forward(torch.tensor(g["xu"]), st.sd, True, ru, dropout=zero)
    np.testing.assert_allclose(np.stack([t.detach().numpy() for t in lab]), g["lab_logits"], atol=2e-5)
    np.testing.assert_allclose(np.stack([t.detach().numpy() for t in un]), g["un_logits"], atol=2e-5)
    cw = float(g["cw"])
    r = O.step_loss(un, lab, torch.tensor(g["yl"]), g["w"], cw, cw)
    np.testing.assert_allclose(float(r["loss"]), g["loss"], rtol=1e-5)
    assert np.array_equal(r["pseudo"].numpy(), g["pseudo"])
    st.opt.zero_grad(); r["loss"].backward()
    for k in st.param_keys:
        ref = g["grad." + k]
        np.testing.assert_allclose(st.sd[k].grad.numpy(), ref, rtol=1e-3, atol=max(1e-7, 1e-5 * np.abs(ref).max()))
    st.opt.step()
    for k in st.sd:
        ref = g["after." + k]
        np.testing.assert_allclose(st.sd[k].detach().numpy(), ref, rtol=0, atol=2.5e-3 if k in st.param_keys else 1e-5)


@pytest.mark.parametrize("C", [4, 2])
def test_sibling_method_terms_g8(C):
    """CCT / UCC / UAMT unsupervised terms: the oracle's restatements against the fixture composed from the
    imported reference callees in the training scripts' order (tools/make_golden.py::g8), values and gradients."""
    g = _load(os.path.join(GOLDEN, "g8_sibling.npz"))
    t = lambda k: torch.tensor(g[k]).requires_grad_(True)
    # CCT
    main, auxs = t(f"cct{C}_main"), [t(f"cct{C}_aux{i}") for i in (1, 2, 3)]
    loss = O.cct_consistency(main, auxs)
    loss.backward()
    np.testing.assert_allclose(float(loss), float(g[f"cct{C}_loss"]), rtol=1e-6)
    np.testing.assert_allclose(main.grad.numpy(), g[f"cct{C}_dmain"], rtol=1e-5, atol=1e-9)
    for i, a in enumerate(auxs, 1):
        np.testing.assert_allclose(a.grad.numpy(), g[f"cct{C}_daux{i}"], rtol=1e-5, atol=1e-9)
    # UCC
    u = {k: t(f"ucc{C}_{k}") for k in ("u1wk", "u2wk", "u1st", "u2st")}
    r = O.ucc_pseudo_supervision(u["u1wk"], u["u2wk"], u["u1st"], u["u2st"])
    r["ps_loss"].backward()
    for k in ("ps_loss", "ps_1_wk", "ps_2_st"):
        np.testing.assert_allclose(float(r[k]), float(g[f"ucc{C}_{k}"]), rtol=1e-6)
    np.testing.assert_allclose(r["variance_1"].detach().numpy(), g[f"ucc{C}_variance_1"], rtol=1e-5, atol=1e-7)
    for k, v in u.items():
        np.testing.assert_allclose(v.grad.numpy(), g[f"ucc{C}_d{k}"], rtol=1e-4, atol=1e-8)
    # UAMT
    student = t(f"uamt{C}_student")
    loss = O.uamt_consistency(student, torch.tensor(g[f"uamt{C}_ema"]), torch.tensor(g[f"uamt{C}_preds"]), float(g[f"uamt{C}_threshold"]))
    loss.backward()
    np.testing.assert_allclose(float(loss), float(g[f"uamt{C}_loss"]), rtol=1e-6)
    np.testing.assert_allclose(student.grad.numpy(), g[f"uamt{C}_dstudent"], rtol=1e-5, atol=1e-9)
