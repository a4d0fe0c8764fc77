"""Unsupervised terms of the comparison methods the reference trains on the same networks and logits beside UAPS
(SURVEY.md section 8, row f-4), on the pair kernels of csrc/pair_ops.hip and the supervised kernels behind
`ce_loss` / `dice_loss` (csrc/loss_sup.hip):

  CCT   CCT/CCT_train.py:195-201      mean squared difference of the main and each auxiliary softmax (both sides trained)
  UCC   UCC/UCC_train.py:213-237      two heads, weak/strong views: KL-uncertainty weighted CE + Dice pseudo-supervision
  UAMT  UAMT/UA_MT_train.py:188-213   softmax-MSE against the EMA teacher, masked by the Monte-Carlo predictive entropy

Per-pixel maps and their gradients are hand-written kernels; the scalar glue between them (means, exp(-v), the mask
threshold) is the handful of torch element-wise / reduction ops the reference's training scripts spell out.  GPU
tensors only: like every op of this package these raise `UapsHipError` on CPU tensors or without the HIP library.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import torch

from . import _lib, consistency
from .losses import ce_loss, dice_loss


def _mse_bwd(a, b, g):
    da = torch.empty_like(a)
    B, Cc, H, W = a.shape
    with _lib.device_guard(a.device):
        rc = _lib.lib().uaps_softmax_mse_bwd(a.data_ptr(), b.data_ptr(), g.data_ptr(), B, Cc, H, W, da.data_ptr(),
                                             _lib.current_stream(a.device))
    _lib.check(rc, "uaps_softmax_mse_bwd")
    return da


class _SoftmaxMseBoth(torch.autograd.Function):
    """(softmax(a) - softmax(b))^2 with the gradient sent to BOTH logit tensors (the map is symmetric, so the target
    side is the same kernel with the roles exchanged)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = consistency._pair(a, b, "softmax_mse_both")
        ctx.save_for_backward(a, b)
        return consistency._fwd(a, b, False, True, False, False)[0]

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da = _mse_bwd(a, b, g) if ctx.needs_input_grad[0] else None
        db = _mse_bwd(b, a, g) if ctx.needs_input_grad[1] else None
        return da, db


class _KlMapBoth(torch.autograd.Function):
    """v[b,h,w] = sum_c KLDivLoss('none')(log_softmax(a), softmax(b)), differentiable w.r.t. both heads."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = consistency._pair(a, b, "kl_map")
        ctx.save_for_backward(a, b)
        return consistency._fwd(a, b, False, False, True, False)[1]

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        B, Cc, H, W = a.shape
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if da is None and db is None:
            return None, None
        with _lib.device_guard(a.device):
            rc = _lib.lib().uaps_softmax_klmap_bwd(a.data_ptr(), b.data_ptr(), g.data_ptr(), B, Cc, H, W,
                                                   da.data_ptr() if da is not None else None,
                                                   db.data_ptr() if db is not None else None, _lib.current_stream(a.device))
        _lib.check(rc, "uaps_softmax_klmap_bwd")
        return da, db


def softmax_mse_both(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """`(torch.softmax(a, 1) - torch.softmax(b, 1)) ** 2` as CCT writes it (CCT_train.py:195): map [B,C,H,W], neither
    side detached."""
    return _SoftmaxMseBoth.apply(a, b)


def kl_map(input_logits: torch.Tensor, target_logits: torch.Tensor) -> torch.Tensor:
    """`torch.sum(nn.KLDivLoss(reduction='none')(log_softmax(input), softmax(target)), dim=1)` (UCC_train.py:213,216):
    map [B,H,W], gradient to both heads."""
    return _KlMapBoth.apply(input_logits, target_logits)


def cct_consistency_loss(main_logits: torch.Tensor, aux_logits: Sequence[torch.Tensor]) -> torch.Tensor:
    """CCT_train.py:195-199: mean over the auxiliary decoders of mean((softmax(main) - softmax(aux_k))^2)."""
    if len(aux_logits) == 0:
        raise ValueError("cct_consistency_loss: at least one auxiliary head expected")
    total = None
    for aux in aux_logits:
        term = torch.mean(softmax_mse_both(main_logits, aux))
        total = term if total is None else total + term
    return total / len(aux_logits)


def ucc_pseudo_supervision(un1_wk: torch.Tensor, un2_wk: torch.Tensor, un1_st: torch.Tensor,
                           un2_st: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """UCC_train.py:213-237, the shipped 'WITH UNCERTAINTY' branch.  un{h}_{wk,st}: logits of head h on the weakly /
    strongly augmented unlabelled batch.  Returns (ps_loss, ps_1_wk, ps_2_st).  `ce_loss` there is
    CrossEntropyLoss() with the default mean reduction, so each bracket is a scalar times the exp(-variance) map."""
    variance_1 = kl_map(un1_wk, un2_st)                                           # :213
    exp_variance_1 = torch.exp(-variance_1)                                       # :214
    variance_2 = kl_map(un1_st, un2_wk)                                           # :216
    exp_variance_2 = torch.exp(-variance_2)                                       # :217
    pseudo_1 = torch.argmax(torch.softmax(un2_wk.detach(), dim=1), dim=1)         # :224
    pseudo_2 = torch.argmax(torch.softmax(un1_wk.detach(), dim=1), dim=1)         # :225
    ps_1_wk = torch.mean(0.5 * (ce_loss(un1_st, pseudo_1) + dice_loss(pseudo_1.unsqueeze(1), un1_st)) * exp_variance_1) \
        + torch.mean(variance_1)                                                  # :231
    ps_2_st = torch.mean(0.5 * (ce_loss(un2_st, pseudo_2) + dice_loss(pseudo_2.unsqueeze(1), un2_st)) * exp_variance_2) \
        + torch.mean(variance_2)                                                  # :232
    return ps_1_wk + ps_2_st, ps_1_wk, ps_2_st                                    # :238


def uamt_threshold(consistency_weight: float) -> float:
    """UA_MT_train.py:212: (0.75 + 2.5 * consistency_weight) * ln 2."""
    return (0.75 + 2.5 * float(consistency_weight)) * math.log(2.0)


def uamt_consistency_loss(student_logits: torch.Tensor, ema_logits: torch.Tensor, mc_mean_probs: torch.Tensor,
                          threshold: float) -> torch.Tensor:
    """UA_MT_train.py:199-214.  mc_mean_probs: the mean over the T noisy teacher passes of softmax(teacher logits)
    (:196-198); its predictive entropy (:199) masks the per-element softmax-MSE between the student and the teacher (:210,
    gradient to the student only), summed and divided by 2 * #unmasked + 1e-16 (:214)."""
    uncertainty = consistency.entropy_map(mc_mean_probs)                          # [B,1,H,W]
    dist = consistency.softmax_mse_loss(student_logits, ema_logits.detach())      # [B,C,H,W]
    mask = (uncertainty < threshold).float()
    return torch.sum(mask * dist) / (2 * torch.sum(mask) + 1e-16)


@torch.no_grad()
def uamt_mc_mean_probs(ema_model, inputs_u: torch.Tensor, num_classes: int, T: int = 8) -> torch.Tensor:
    """UA_MT_train.py:188-198: T noisy forward passes of the EMA teacher (two stacked copies of the batch per call,
    Gaussian noise 0.1 clamped to +-0.2), softmax, mean over the passes -> [B,C,H,W]."""
    B, _, h, w = inputs_u.shape
    u_batch_r = inputs_u.repeat(2, 1, 1, 1)
    preds = torch.zeros([B * T, num_classes, h, w], device=inputs_u.device)
    for i in range(T // 2):
        ema_inputs = u_batch_r + torch.clamp(torch.randn_like(u_batch_r) * 0.1, -0.2, 0.2)
        out = ema_model(ema_inputs)
        out = out[0] if isinstance(out, (tuple, list)) else out
        preds[2 * B * i:2 * B * (i + 1)] = out
    preds = torch.softmax(preds, dim=1).reshape(T, B, num_classes, h, w)
    return torch.mean(preds, dim=0)
