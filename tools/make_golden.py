#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference, read-only). The GPU box
never sees the reference; it only sees the .npz files this script wrote.

What is "the reference" here:
  * `UNet_UAPS`, `ConvBlock`, `DownBlock`, `UpBlock`, `Encoder`, `Decoder`, `FeatureNoise`,
    `FeatureDropout`, `Dropout`   -- utilities/UAPS_unet.py
  * `dice_loss`                   -- utilities/pytorch_losses.py:54-89
  * `sigmoid_rampup`              -- utilities/ramps.py:19-26
  * `mIoU`, `mDice`, `pixel_accuracy` -- utilities/metrics.py:8-61
  * `softmax_kl_loss`, `softmax_mse_loss`, `entropy_map` -- utilities/losses_1.py
  * `kl_loss`                     -- utilities/losses_2.py:201-213
are the real imported objects.  UAPS_train.py itself cannot be imported (tensorboardX,
cv2, albumentations, dataset walk at import), so its step, lines 177-282, is composed
below from those imported pieces plus the same torch builtins it instantiates at
lines 73-75, in the same order and association, generalised over the number of heads.

Usage: python tools/make_golden.py            (writes tests/golden/*.npz)
"""
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn as nn

import utilities.UAPS_unet as R_unet                      # noqa: E402
from utilities.pytorch_losses import dice_loss as R_dice  # noqa: E402
from utilities.ramps import sigmoid_rampup as R_ramp      # noqa: E402
from utilities import metrics as R_metrics                # noqa: E402
import utilities.losses_1 as R_l1                         # noqa: E402
import utilities.losses_2 as R_l2                         # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
os.makedirs(OUT, exist_ok=True)

# the loss objects of UAPS_train.py:73-75
kl_distance = nn.KLDivLoss(reduction="none")
log_sm = nn.LogSoftmax(dim=1)
ce_loss = nn.CrossEntropyLoss()


def ref_step_losses(un_logits, lab_logits, labels, w, cw1, cw2):
    """UAPS_train.py:186-282 for D heads (D=4 is the literal reference)."""
    D = len(un_logits)
    un_soft = [torch.softmax(z, dim=1) for z in un_logits]                       # :186-189
    # supervised branch :194-218
    ce = [ce_loss(o, labels.long()) for o in lab_logits]
    dc = [R_dice(labels.unsqueeze(1), o) for o in lab_logits]
    per_head = [0.5 * (a + b) for a, b in zip(ce, dc)]
    sup = per_head[0]
    for t in per_head[1:]:
        sup = sup + t
    sup = sup / D
    # :223
    preds = un_soft[0]
    for s in un_soft[1:]:
        preds = preds + s
    preds = preds / D
    var = [torch.sum(kl_distance(log_sm(z), preds), dim=1) for z in un_logits]   # :226-235
    evar = [torch.exp(-v) for v in var]
    ave = var[0]
    for v in var[1:]:
        ave = ave + v
    ave = ave / D
    l_uncert = torch.mean(ave)                                                   # :241-243
    mixed = w[0] * un_soft[0].detach()                                            # :252-255
    for k in range(1, D):
        mixed = mixed + w[k] * un_soft[k].detach()
    pseudo = torch.argmax(mixed, dim=1, keepdim=False)
    ps = [0.5 * (ce_loss(z, pseudo) + R_dice(pseudo.unsqueeze(1), z)) for z in un_logits]   # :259-262
    ce_ps = [ce_loss(z, pseudo) for z in un_logits]
    dice_ps = [R_dice(pseudo.unsqueeze(1), z) for z in un_logits]
    psl = [torch.mean(p * e) for p, e in zip(ps, evar)]                           # :265-268
    ps_loss = psl[0]
    for t in psl[1:]:
        ps_loss = ps_loss + t
    ps_loss = ps_loss / D                                                         # :277
    loss = sup + cw1 * ps_loss + cw2 * l_uncert                                   # :282
    return dict(un_soft=un_soft, preds=preds, var=var, evar=evar, l_uncert=l_uncert, mixed=mixed,
                pseudo=pseudo, ce_ps=ce_ps, dice_ps=dice_ps, ps=ps, psl=psl, ps_loss=ps_loss,
                sup=sup, ce_sup=ce, dice_sup=dc, loss=loss)


def rect_labels(rng, B, C, H, W):
    y = np.zeros((B, H, W), np.int64)
    for b in range(B):
        for _ in range(rng.integers(1, 4)):
            c = rng.integers(1, C) if C > 1 else 0
            h0, w0 = rng.integers(0, H), rng.integers(0, W)
            h1, w1 = min(H, h0 + rng.integers(1, max(2, H // 2))), min(W, w0 + rng.integers(1, max(2, W // 2)))
            y[b, h0:h1, w0:w1] = c
    return y


def g1_case(name, D, B, C, H, W, seed, scale=2.0, neartie=False, uniform_labels=False):
    rng = np.random.default_rng(seed)
    un = [(rng.standard_normal((B, C, H, W)) * scale).astype(np.float32) for _ in range(D)]
    lab = [(rng.standard_normal((B, C, H, W)) * scale).astype(np.float32) for _ in range(D)]
    if neartie:
        # make every head agree on two classes being (almost) tied on half of the pixels
        for k in range(D):
            un[k][:, 1, :, : W // 2] = un[k][:, 0, :, : W // 2] + np.float32(1e-6) * rng.standard_normal((B, H, W // 2)).astype(np.float32)
    labels = rng.integers(0, C, (B, H, W)).astype(np.int64) if uniform_labels else rect_labels(rng, B, C, H, W)
    w = rng.dirichlet(np.ones(D), size=1)[0]                     # float64, as np.random.dirichlet at :251
    out = {"D": D, "B": B, "C": C, "H": H, "W": W, "w": w, "labels": labels,
           "un_logits": np.stack(un), "lab_logits": np.stack(lab)}
    ramp0 = R_ramp(0, 200)
    for tag, (cw1, cw2) in {"r0": (0.1 * ramp0, 0.1 * ramp0), "full": (0.1, 0.1), "mt": (0.1, 1.0)}.items():
        un_t = [torch.tensor(a, requires_grad=True) for a in un]
        lab_t = [torch.tensor(a, requires_grad=True) for a in lab]
        r = ref_step_losses(un_t, lab_t, torch.tensor(labels), w, cw1, cw2)
        r["loss"].backward()
        out[f"cw_{tag}"] = np.array([cw1, cw2], np.float64)
        out[f"loss_{tag}"] = r["loss"].detach().numpy()
        out[f"g_un_{tag}"] = np.stack([t.grad.numpy() for t in un_t])
        out[f"g_lab_{tag}"] = np.stack([t.grad.numpy() for t in lab_t])
        if tag == "full":
            out["un_soft"] = np.stack([t.detach().numpy() for t in r["un_soft"]])
            out["preds"] = r["preds"].detach().numpy()
            out["var"] = np.stack([t.detach().numpy() for t in r["var"]])
            out["evar"] = np.stack([t.detach().numpy() for t in r["evar"]])
            out["mixed"] = r["mixed"].numpy()
            out["pseudo"] = r["pseudo"].numpy()
            for key in ("ce_ps", "dice_ps", "ps", "psl", "ce_sup", "dice_sup"):
                out[key] = np.array([float(t) for t in r[key]], np.float32)
            for key in ("l_uncert", "ps_loss", "sup"):
                out[key] = np.float32(float(r[key]))
    np.savez_compressed(os.path.join(OUT, f"g1_{name}.npz"), **out)
    print("g1", name, {k: out[k] for k in ("loss_full", "sup", "ps_loss", "l_uncert")})


def g2():
    out = {}
    ts = [0, 1, 50, 199, 200, 250, -3, 37.5]
    Rs = [200, 150, 0]
    out["ramp_t"] = np.array(ts, np.float64)
    out["ramp_R"] = np.array(Rs, np.float64)
    out["ramp"] = np.array([[R_ramp(t, R) for R in Rs] for t in ts], np.float64)
    rng = np.random.default_rng(7)
    for C in (4, 7, 2):
        a = torch.tensor((rng.standard_normal((2, C, 12, 10)) * 2).astype(np.float32))
        b = torch.tensor((rng.standard_normal((2, C, 12, 10)) * 2).astype(np.float32))
        y = torch.tensor(rng.integers(0, C, (2, 12, 10)).astype(np.int64))
        out[f"a{C}"], out[f"b{C}"], out[f"y{C}"] = a.numpy(), b.numpy(), y.numpy()
        out[f"dice{C}"] = np.float32(R_dice(y.unsqueeze(1), a))
        out[f"dice{C}_eps"] = np.float32(R_dice(y.unsqueeze(1), a, eps=1e-3))
        out[f"ce{C}"] = np.float32(ce_loss(a, y))
        out[f"softmax_kl{C}"] = np.float32(R_l1.softmax_kl_loss(a, b))
        out[f"softmax_mse{C}"] = R_l1.softmax_mse_loss(a, b).numpy()
        p = torch.softmax(a, dim=1)
        q = torch.softmax(b, dim=1)
        out[f"entropy_map{C}"] = R_l1.entropy_map(p).numpy()
        out[f"entropy_min{C}"] = np.float32(R_l1.entropy_minmization(p))
        out[f"kl_loss{C}"] = np.float32(R_l2.kl_loss(p, q))
    np.savez_compressed(os.path.join(OUT, "g2_losses.npz"), **out)
    print("g2 ramp row t=50:", out["ramp"][2])


def g3():
    """Perturbations with recorded randomness (UAPS_unet.py:156-185)."""
    out = {}
    rng = np.random.default_rng(11)
    shapes = [(2, 4, 8, 8), (3, 6, 5, 7), (2, 16, 4, 4)]
    for i, shp in enumerate(shapes):
        x = torch.tensor(rng.standard_normal(shp).astype(np.float32))
        out[f"x{i}"] = x.numpy()
        # FeatureNoise: capture the noise by re-seeding the torch generator it samples from.
        torch.manual_seed(100 + i)
        fn = R_unet.FeatureNoise()
        y = fn(x)
        torch.manual_seed(100 + i)
        noise = fn.uni_dist.sample(x.shape[1:])
        assert torch.equal(y, x.mul(noise.unsqueeze(0)) + x)
        out[f"noise{i}"], out[f"noise_y{i}"] = noise.numpy(), y.numpy()
        # FeatureDropout: numpy global RNG scalar.
        np.random.seed(200 + i)
        yd = R_unet.FeatureDropout(x)
        np.random.seed(200 + i)
        u = np.random.uniform(0.7, 0.9)
        out[f"fd_u{i}"], out[f"fd_y{i}"] = np.float64(u), yd.numpy()
        # Dropout(x, 0.5), always training
        torch.manual_seed(300 + i)
        yb = R_unet.Dropout(x)
        out[f"bern_y{i}"] = yb.numpy()
        out[f"bern_mask{i}"] = (yb != 0).numpy().astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "g3_perturb.npz"), **out)
    print("g3 done")


def sd_np(m, prefix=""):
    return {prefix + k: v.detach().numpy().copy() for k, v in m.state_dict().items()}


def g4():
    out = {}
    # (i) state_dict key/shape/dtype lists of the real-width nets
    for in_chns, ncls in ((3, 4), (1, 7), (3, 2), (3, 6)):
        torch.manual_seed(0)
        net = R_unet.UNet_UAPS(in_chns, ncls)
        keys = list(net.state_dict().keys())
        out[f"keys_{in_chns}_{ncls}"] = np.array(keys)
        out[f"shapes_{in_chns}_{ncls}"] = np.array([",".join(map(str, v.shape)) for v in net.state_dict().values()])
        out[f"dtypes_{in_chns}_{ncls}"] = np.array([str(v.dtype) for v in net.state_dict().values()])
        out[f"param_names_{in_chns}_{ncls}"] = np.array([n for n, _ in net.named_parameters()])
        out[f"nparams_{in_chns}_{ncls}"] = np.int64(sum(p.numel() for p in net.parameters()))
    torch.manual_seed(0)
    un = R_unet.UNet(3, 4)
    out["keys_unet_3_4"] = np.array(list(un.state_dict().keys()))
    # (ii) block-level numerics, small channel counts
    rng = np.random.default_rng(21)
    torch.manual_seed(1)
    cb = R_unet.ConvBlock(3, 4, 0.0)
    x = torch.tensor(rng.standard_normal((2, 3, 10, 12)).astype(np.float32))
    out.update(sd_np(cb, "cb."))
    out["cb_x"] = x.numpy()
    cb.train(); out["cb_y_train"] = cb(x).detach().numpy()
    out.update(sd_np(cb, "cb_after."))       # running stats after one train forward
    cb.eval(); out["cb_y_eval"] = cb(x).detach().numpy()
    torch.manual_seed(2)
    db = R_unet.DownBlock(4, 6, 0.0)
    x = torch.tensor(rng.standard_normal((2, 4, 12, 8)).astype(np.float32))
    out.update(sd_np(db, "db.")); out["db_x"] = x.numpy()
    db.train(); out["db_y_train"] = db(x).detach().numpy()
    db.eval(); out["db_y_eval"] = db(x).detach().numpy()
    torch.manual_seed(3)
    ub = R_unet.UpBlock(8, 4, 4, 0.0)
    x1 = torch.tensor(rng.standard_normal((2, 8, 5, 6)).astype(np.float32))
    x2 = torch.tensor(rng.standard_normal((2, 4, 10, 12)).astype(np.float32))
    out.update(sd_np(ub, "ub.")); out["ub_x1"], out["ub_x2"] = x1.numpy(), x2.numpy()
    ub.train(); out["ub_y_train"] = ub(x1, x2).detach().numpy()
    ub.eval(); out["ub_y_eval"] = ub(x1, x2).detach().numpy()
    # (iii) narrow whole net from the reference Encoder/Decoder classes
    params = {"in_chns": 3, "feature_chns": [2, 4, 8, 16, 32], "dropout": [0.05, 0.1, 0.2, 0.3, 0.5],
              "class_num": 4, "bilinear": False, "acti_func": "relu"}
    torch.manual_seed(4)
    enc, dec = R_unet.Encoder(params), R_unet.Decoder(params)
    x = torch.tensor(rng.standard_normal((2, 3, 32, 32)).astype(np.float32))
    # a train-mode forward first so BN running stats are non-trivial, then eval output
    enc.train(); dec.train()
    with torch.no_grad():
        dec(enc(x))
    enc.eval(); dec.eval()
    out.update(sd_np(enc, "narrow.encoder.")); out.update(sd_np(dec, "narrow.main_decoder."))
    out["narrow_x"] = x.numpy()
    with torch.no_grad():
        feats = enc(x)
        out["narrow_y_eval"] = dec(feats).numpy()
        for i, f in enumerate(feats):
            out[f"narrow_feat{i}"] = f.numpy()
    np.savez_compressed(os.path.join(OUT, "g4_model.npz"), **out)
    print("g4 nparams(3,4) =", out["nparams_3_4"], "nkeys", len(out["keys_3_4"]))


def g5():
    out = {}
    rng = np.random.default_rng(31)
    cases = []
    for i in range(4):
        lg = torch.tensor((rng.standard_normal((2, 4, 16, 16)) * 2).astype(np.float32))
        y = torch.tensor(rng.integers(0, 4, (2, 16, 16)).astype(np.int64))
        if i == 1:
            y[y == 2] = 0            # absent class -> NaN-mean path
        if i == 2:
            y[:] = 0                  # all classes 1..3 absent -> nanmean of all-NaN
        if i == 3:
            lg[:, 3] = -50.0          # class never predicted
        cases.append((lg, y))
    import warnings
    for i, (lg, y) in enumerate(cases):
        out[f"logits{i}"], out[f"labels{i}"] = lg.numpy(), y.numpy()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out[f"miou{i}"] = np.float64(R_metrics.mIoU(lg, y))
            out[f"mdice{i}"] = np.float64(R_metrics.mDice(lg, y))
        out[f"acc{i}"] = np.float64(R_metrics.pixel_accuracy(lg, y))
    np.savez_compressed(os.path.join(OUT, "g5_metrics.npz"), **out)
    print("g5", [float(out[f"miou{i}"]) for i in range(4)])


class _FixedRandomness:
    """Makes UNet_UAPS.forward deterministic by recording/replaying its three random draws.

    FeatureNoise samples from the torch CPU generator, Dropout from the torch generator,
    FeatureDropout from numpy's global RNG (UAPS_unet.py:156-185); nn.Dropout inside the
    encoder ConvBlocks also uses the torch generator.  We seed both generators right before
    the forward, so replay = same seeds.  The *recorded* tensors (noise, masks, thresholds)
    are captured with forward hooks / wrappers so that the build can inject them.
    """


def g6():
    """One full restated step on a narrow UNet_UAPS-shaped net (B=2, 32x32), eval-free.

    Randomness is removed instead of recorded: encoder dropout p=0 and the three feature
    perturbations replaced by recorded tensors applied with the reference formulas
    (x*n+x ; x*mask*2 ; x*(mean<thr)).  The perturbation functions themselves are pinned by g3.
    """
    out = {}
    rng = np.random.default_rng(41)
    params = {"in_chns": 3, "feature_chns": [2, 4, 8, 16, 32], "dropout": [0.0] * 5,
              "class_num": 4, "bilinear": False, "acti_func": "relu"}
    torch.manual_seed(5)
    enc = R_unet.Encoder(params)
    decs = [R_unet.Decoder(params) for _ in range(4)]
    names = ["main_decoder", "aux_decoder1", "aux_decoder2", "aux_decoder3"]
    mods = nn.ModuleDict({"encoder": enc, **{n: d for n, d in zip(names, decs)}})
    for k, v in mods.state_dict().items():
        out["init." + k] = v.numpy().copy()
    B, H, W = 2, 32, 32
    xl = torch.tensor(rng.standard_normal((B, 3, H, W)).astype(np.float32))
    xu = torch.tensor(rng.standard_normal((B, 3, H, W)).astype(np.float32))
    yl = torch.tensor(rect_labels(rng, B, 4, H, W))
    w = rng.dirichlet(np.ones(4), size=1)[0]
    out.update(xl=xl.numpy(), xu=xu.numpy(), yl=yl.numpy(), w=w)
    fshapes = [(2, 32, 32), (4, 16, 16), (8, 8, 8), (16, 4, 4), (32, 2, 2)]
    rec = {}
    for tag in ("l", "u"):
        rec[tag] = {
            "noise": [torch.tensor(rng.uniform(-0.3, 0.3, s).astype(np.float32)) for s in fshapes],
            "mask": [torch.tensor((rng.random((B,) + s) < 0.5).astype(np.float32)) for s in fshapes],
            "u": [float(rng.uniform(0.7, 0.9)) for _ in fshapes],
        }
        for i in range(5):
            out[f"noise_{tag}{i}"] = rec[tag]["noise"][i].numpy()
            out[f"mask_{tag}{i}"] = rec[tag]["mask"][i].numpy().astype(np.uint8)
            out[f"u_{tag}{i}"] = np.float64(rec[tag]["u"][i])

    def fdrop(x, u):      # UAPS_unet.py:161-169 with the numpy draw replaced by the recorded u
        attention = torch.mean(x, dim=1, keepdim=True)
        max_val, _ = torch.max(attention.view(x.size(0), -1), dim=1, keepdim=True)
        threshold = max_val * u
        threshold = threshold.view(x.size(0), 1, 1, 1).expand_as(attention)
        return x.mul((attention < threshold).float())

    def forward(x, r):    # UAPS_unet.py:224-233
        f = enc(x)
        o0 = decs[0](f)
        o1 = decs[1]([t.mul(n.unsqueeze(0)) + t for t, n in zip(f, r["noise"])])
        o2 = decs[2]([t * m * 2.0 for t, m in zip(f, r["mask"])])       # F.dropout(p=.5): mask/(1-p)
        o3 = decs[3]([fdrop(t, u) for t, u in zip(f, r["u"])])
        return [o0, o1, o2, o3]

    mods.train()
    opt = torch.optim.Adam(mods.parameters(), lr=1e-3)
    lab_logits = forward(xl, rec["l"])
    un_logits = forward(xu, rec["u"])
    cw = 0.1 * R_ramp(3, 200)
    r = ref_step_losses(un_logits, lab_logits, yl, w, cw, cw)
    opt.zero_grad()
    r["loss"].backward()
    out["cw"] = np.float64(cw)
    out["loss"] = r["loss"].detach().numpy()
    out["sup"] = r["sup"].detach().numpy()
    out["ps_loss"] = r["ps_loss"].detach().numpy()
    out["l_uncert"] = r["l_uncert"].detach().numpy()
    out["pseudo"] = r["pseudo"].numpy()
    out["lab_logits"] = np.stack([t.detach().numpy() for t in lab_logits])
    out["un_logits"] = np.stack([t.detach().numpy() for t in un_logits])
    for n, p in mods.named_parameters():
        out["grad." + n] = p.grad.numpy().copy()
    opt.step()
    for k, v in mods.state_dict().items():
        out["after." + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g6_step.npz"), **out)
    print("g6 loss", float(out["loss"]), "sup", float(out["sup"]))


def g7():
    """ResNet-50 backbone of utilities/resnet.py (dilated layer3/layer4): outputs of base_forward for RNG-free formula
    weights (tests/formula_weights.py) in eval and train mode, plus the ordered state_dict keys and shapes."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from formula_weights import formula_state_dict, formula_input
    import utilities.resnet as R_res
    torch.manual_seed(0)
    net = R_res.resnet50()
    net.load_state_dict(formula_state_dict(net.state_dict()))
    x = formula_input((2, 3, 48, 48))
    out = {"x": x.numpy(), "keys": np.array(list(net.state_dict().keys())),
           "shapes": np.array([str(tuple(v.shape)) for v in net.state_dict().values()])}
    net.eval()
    with torch.no_grad():
        for i, c in enumerate(net.base_forward(x)):
            out[f"eval_c{i + 1}"] = c.numpy()
    net.train()
    with torch.no_grad():
        cs = net.base_forward(x)
    out["train_c1"], out["train_c4"] = cs[0].numpy(), cs[3].numpy()
    out["train_rm_layer4_2_bn3"] = net.layer4[2].bn3.running_mean.numpy()
    np.savez_compressed(os.path.join(OUT, "g7_resnet.npz"), **out)


def g8():
    """Unsupervised terms of the sibling methods: CCT (CCT/CCT_train.py:195-199), UCC
    (UCC/UCC_train.py:213-238, the shipped 'WITH UNCERTAINTY' branch) and UAMT (UAMT/UA_MT_train.py:199, 210-214),
    composed in the order of the training scripts from the imported reference callees (dice_loss, softmax_mse_loss) and
    the torch modules the scripts instantiate (:67-69 of UCC_train.py); the scripts themselves import cv2 /
    tensorboardX and cannot be imported here.  Values and the gradients w.r.t. every logit tensor."""
    out = {}
    rng = np.random.default_rng(23)
    kl_distance = nn.KLDivLoss(reduction='none')
    log_sm = torch.nn.LogSoftmax(dim=1)
    for C in (4, 2):
        B, H, W = 2, 12, 10

        def t(grad=True):
            return torch.tensor((rng.standard_normal((B, C, H, W)) * 2).astype(np.float32), requires_grad=grad)

        # ---- CCT
        un_outputs, a1, a2, a3 = t(), t(), t(), t()
        un_outputs_soft = torch.softmax(un_outputs, dim=1)
        c1 = torch.mean((un_outputs_soft - torch.softmax(a1, dim=1)) ** 2)
        c2 = torch.mean((un_outputs_soft - torch.softmax(a2, dim=1)) ** 2)
        c3 = torch.mean((un_outputs_soft - torch.softmax(a3, dim=1)) ** 2)
        consistency_loss = (c1 + c2 + c3) / 3
        consistency_loss.backward()
        for k, v in (("main", un_outputs), ("aux1", a1), ("aux2", a2), ("aux3", a3)):
            out[f"cct{C}_{k}"] = v.detach().numpy(); out[f"cct{C}_d{k}"] = v.grad.numpy()
        out[f"cct{C}_loss"] = np.float32(consistency_loss.item())

        # ---- UCC
        un_outputs1_wk, un_outputs2_wk, un_outputs1_st, un_outputs2_st = t(), t(), t(), t()
        un_outputs1_wk_soft = torch.softmax(un_outputs1_wk, dim=1)
        un_outputs2_wk_soft = torch.softmax(un_outputs2_wk, dim=1)
        un_outputs2_st_soft = torch.softmax(un_outputs2_st, dim=1)
        variance_1 = torch.sum(kl_distance(log_sm(un_outputs1_wk), un_outputs2_st_soft), dim=1)
        exp_variance_1 = torch.exp(-variance_1)
        variance_2 = torch.sum(kl_distance(log_sm(un_outputs1_st), un_outputs2_wk_soft), dim=1)
        exp_variance_2 = torch.exp(-variance_2)
        pseudo_1 = torch.argmax(un_outputs2_wk_soft.detach(), dim=1, keepdim=False)
        pseudo_2 = torch.argmax(un_outputs1_wk_soft.detach(), dim=1, keepdim=False)
        ps_1_wk = torch.mean(0.5 * (ce_loss(un_outputs1_st, pseudo_1) + R_dice(pseudo_1.unsqueeze(1), un_outputs1_st)) * exp_variance_1) + torch.mean(variance_1)
        ps_2_st = torch.mean(0.5 * (ce_loss(un_outputs2_st, pseudo_2) + R_dice(pseudo_2.unsqueeze(1), un_outputs2_st)) * exp_variance_2) + torch.mean(variance_2)
        ps_loss = ps_1_wk + ps_2_st
        ps_loss.backward()
        for k, v in (("u1wk", un_outputs1_wk), ("u2wk", un_outputs2_wk), ("u1st", un_outputs1_st), ("u2st", un_outputs2_st)):
            out[f"ucc{C}_{k}"] = v.detach().numpy(); out[f"ucc{C}_d{k}"] = v.grad.numpy()
        out[f"ucc{C}_ps_loss"] = np.float32(ps_loss.item()); out[f"ucc{C}_ps_1_wk"] = np.float32(ps_1_wk.item()); out[f"ucc{C}_ps_2_st"] = np.float32(ps_2_st.item())
        out[f"ucc{C}_variance_1"] = variance_1.detach().numpy(); out[f"ucc{C}_variance_2"] = variance_2.detach().numpy()

        # ---- UAMT (T = 4 stand-in teacher passes; the real script uses 8 noisy passes of the EMA model)
        un_outputs_1, ema_output = t(), t(False)
        T = 4
        preds = torch.softmax(torch.stack([t(False) for _ in range(T)]), dim=2)
        preds = torch.mean(preds, dim=0)
        uncertainty = -1.0 * torch.sum(preds * torch.log(preds + 1e-6), dim=1, keepdim=True)
        consistency_weight = {4: 0.3, 2: 0.0}[C]      # puts the threshold inside the range of the entropies (a mixed mask)
        consistency_dist = R_l1.softmax_mse_loss(un_outputs_1, ema_output)
        threshold = (0.75 + 2.5 * consistency_weight) * np.log(2)
        mask = (uncertainty < threshold).float()
        uamt_loss = torch.sum(mask * consistency_dist) / (2 * torch.sum(mask) + 1e-16)
        uamt_loss.backward()
        out[f"uamt{C}_student"] = un_outputs_1.detach().numpy(); out[f"uamt{C}_ema"] = ema_output.numpy()
        out[f"uamt{C}_preds"] = preds.numpy(); out[f"uamt{C}_threshold"] = np.float64(threshold)
        out[f"uamt{C}_mask_frac"] = np.float32(mask.mean().item()); out[f"uamt{C}_loss"] = np.float32(uamt_loss.item())
        out[f"uamt{C}_dstudent"] = un_outputs_1.grad.numpy()
        print(f"g8 C={C}: cct {float(consistency_loss):.6f} ucc {float(ps_loss):.6f} uamt {float(uamt_loss):.6f} mask {float(mask.mean()):.3f}")
    np.savez_compressed(os.path.join(OUT, "g8_sibling.npz"), **out)


def main():
    torch.set_num_threads(4)
    g1_case("neu", 4, 2, 4, 16, 16, seed=0)
    g1_case("dagm7", 4, 2, 7, 12, 12, seed=1)
    g1_case("kosdd2", 4, 1, 2, 16, 32, seed=2)
    g1_case("d6c2", 6, 2, 2, 16, 16, seed=3)
    g1_case("d2", 2, 2, 4, 8, 8, seed=4)
    g1_case("d1", 1, 2, 4, 16, 16, seed=5)
    g1_case("ragged", 4, 3, 4, 5, 7, seed=6, uniform_labels=True)
    g1_case("saturated", 4, 2, 4, 16, 16, seed=7, scale=60.0)
    g1_case("neartie", 4, 2, 4, 16, 16, seed=8, neartie=True)
    g1_case("d8c8", 8, 1, 8, 6, 10, seed=9, uniform_labels=True)
    g2()
    g3()
    g4()
    g5()
    g6()
    g7()
    g8()


if __name__ == "__main__":
    main()
