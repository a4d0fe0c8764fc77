"""Host side of the fused UAPS loss block (HIP kernels in csrc/loss_*.hip, C ABI include/uaps_hip.h).

Mirrors what the reference step computes between the model forward and `loss.backward()`
(UAPS_train.py:186-282) and the helper it calls (utilities/pytorch_losses.py:54-89 `dice_loss`).
Every function here launches hand-written kernels on the current HIP stream; nothing falls back to
PyTorch ops or to the CPU.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Dict, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, config

MAX_HEADS, MAX_CLASSES = 8, 8


# ---- offsets into the `scalars` outputs (include/uaps_hip.h) ----------------------------------
def _u_off(D, C):
    b = 4 * D + 4
    return {"ce": 0, "dice": D, "s": 2 * D, "E": 3 * D, "ps_loss": 4 * D, "l_uncert": 4 * D + 1, "loss": 4 * D + 2, "total": 4 * D + 3,
            "a1": b, "a2": b + D * C, "I": b + 2 * D * C, "card": b + 3 * D * C, "cnt": b + 4 * D * C,
            "n": b + 4 * D * C + C}


def _s_off(D, C):
    b = 2 * D + 2
    return {"ce": 0, "dice": D, "sup": 2 * D, "bad": 2 * D + 1, "a1": b, "a2": b + D * C, "I": b + 2 * D * C,
            "card": b + 3 * D * C, "cnt": b + 4 * D * C, "n": b + 4 * D * C + C}


_ws_cache: Dict[Tuple[int, int], torch.Tensor] = {}

# bench.py sets this to a dict {kernel name: [(start_event, end_event), ...]} to time individual launches with HIP events
# attached to the main kernel's dispatch (_lib.LaunchTimer: the pair-loss entries time pair_fwd_kernel / pair_bwd_kernel, the
# separate-branch entries fall back to event-to-event); None (the default) records nothing.
KERNEL_EVENTS: Optional[Dict[str, list]] = None


# tuning knob of the pair-loss kernels (cap on the blocks per branch; 0 = one group of pixels per thread)
PAIR_CFG = config.integer("UAPS_PAIR_CFG", 0)


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.t = _lib.LaunchTimer().__enter__() if KERNEL_EVENTS is not None else None
        return self

    def __exit__(self, *a):
        if self.t is not None:
            self.t.__exit__()
            if KERNEL_EVENTS is not None:
                KERNEL_EVENTS.setdefault(self.name, []).append((self.t.start, self.t.stop))
        return False


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Per (device, stream) scratch for the block partial sums; grown on demand, never shrunk."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _lib.current_stream(device))
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def _check_heads(logits: Sequence[torch.Tensor], what: str):
    D = len(logits)
    if not 1 <= D <= MAX_HEADS:
        raise ValueError(f"{what}: {D} heads, supported 1..{MAX_HEADS}")
    z0 = logits[0]
    _lib.require_device(z0, what)
    if z0.dim() != 4:
        raise ValueError(f"{what}: logits must be [B,C,H,W], got {tuple(z0.shape)}")
    B, Cc, H, W = z0.shape
    if not 2 <= Cc <= MAX_CLASSES:
        raise ValueError(f"{what}: {Cc} classes, supported 2..{MAX_CLASSES}")
    out = []
    for z in logits:
        if z.shape != z0.shape or z.device != z0.device:
            raise ValueError(f"{what}: all heads must share shape and device")
        if z.dtype != torch.float32:
            raise TypeError(f"{what}: logits must be float32 (the reference computes the loss in fp32), got {z.dtype}")
        out.append(z.contiguous())
    return out, D, B, Cc, H, W


class UnsupOut(NamedTuple):
    loss: torch.Tensor          # cw1 * ps_loss + cw2 * l_uncert   (differentiable, 0-dim)
    pseudo: torch.Tensor        # int64 [B,H,W]
    var: Optional[torch.Tensor]  # fp32 [D,B,H,W] KL(mean || p_k), None if not requested
    scalars: torch.Tensor       # fp32 vector, see scalar_views()


class _UnsupLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, cw1, cw2, eps, want_var, *logits):
        ctx.set_materialize_grads(False)
        zs, D, B, Cc, H, W = _check_heads(logits, "uaps_unsup_loss")
        dev = zs[0].device
        L = _lib.lib()
        off = _u_off(D, Cc)
        need = C.c_size_t()
        _lib.check(L.uaps_loss_workspace_bytes(D, B, Cc, H, W, C.byref(need)), "uaps_loss_workspace_bytes")
        ws = _workspace(dev, need.value)
        pseudo = torch.empty((B, H, W), dtype=torch.int64, device=dev)
        var = torch.empty((D, B, H, W), dtype=torch.float32, device=dev) if want_var else None
        scalars = torch.empty(off["n"], dtype=torch.float32, device=dev)
        w64 = (C.c_double * D)(*[float(x) for x in w])
        with _lib.device_guard(dev), _timed("uaps_unsup_fwd"):
            rc = L.uaps_unsup_fwd(_lib.ptr_array(zs), w64, D, B, Cc, H, W, float(cw1), float(cw2), float(eps),
                                  pseudo.data_ptr(), var.data_ptr() if want_var else None, scalars.data_ptr(),
                                  ws.data_ptr(), ws.numel(), _lib.current_stream(dev))
        _lib.check(rc, "uaps_unsup_fwd")
        ctx.save_for_backward(pseudo, scalars, *zs)
        ctx.meta = (D, B, Cc, H, W, float(cw1), float(cw2))
        loss = scalars[off["loss"]].clone()
        if want_var:
            ctx.mark_non_differentiable(pseudo, var, scalars)
            return loss, pseudo, var, scalars
        ctx.mark_non_differentiable(pseudo, scalars)
        return loss, pseudo, scalars

    @staticmethod
    def backward(ctx, g_loss, *unused):
        pseudo, scalars, *zs = ctx.saved_tensors
        D, B, Cc, H, W, cw1, cw2 = ctx.meta
        if g_loss is None:
            return (None,) * (5 + D)
        dev = zs[0].device
        g = g_loss.contiguous().to(torch.float32)
        dz = [torch.empty_like(z) for z in zs]
        with _lib.device_guard(dev), _timed("uaps_unsup_bwd"):
            rc = _lib.lib().uaps_unsup_bwd(_lib.ptr_array(zs), pseudo.data_ptr(), scalars.data_ptr(), cw1, cw2,
                                           g.data_ptr(), D, B, Cc, H, W, _lib.ptr_array(dz), _lib.current_stream(dev))
        _lib.check(rc, "uaps_unsup_bwd")
        return (None, None, None, None, None) + tuple(dz)


def uaps_unsup_loss(un_logits: Sequence[torch.Tensor], w, cw1: float, cw2: float, eps: float = 1e-7,
                    return_var: bool = False) -> UnsupOut:
    """Unsupervised branch of the step, UAPS_train.py:186-189 + 223-282, as two kernel launches.

    un_logits : the D decoder outputs on the unlabelled batch, main head first (UAPS_unet.py:233)
    w         : D mixing weights, e.g. np.random.dirichlet(np.ones(D)) (UAPS_train.py:251)
    cw1, cw2  : consistency weights (UAPS_train.py:279-280)
    Returns loss = cw1*ps_loss + cw2*l_uncert (differentiable w.r.t. every head), the arg-max
    pseudo-label, optionally the D uncertainty maps, and the scalar block (see unsup_scalars()).
    """
    if len(w) != len(un_logits):
        raise ValueError("one mixing weight per head")
    out = _UnsupLoss.apply(tuple(float(x) for x in w), cw1, cw2, eps, bool(return_var), *un_logits)
    if return_var:
        return UnsupOut(out[0], out[1], out[2], out[3])
    return UnsupOut(out[0], out[1], None, out[2])


def unsup_scalars(scalars: torch.Tensor, D: int, C_: int) -> Dict[str, torch.Tensor]:
    """Named views into the unsupervised scalar block (no copies, no sync)."""
    o = _u_off(D, C_)
    return {"ce": scalars[o["ce"]:o["ce"] + D], "dice": scalars[o["dice"]:o["dice"] + D], "s": scalars[o["s"]:o["s"] + D],
            "E": scalars[o["E"]:o["E"] + D], "ps_loss": scalars[o["ps_loss"]], "l_uncert": scalars[o["l_uncert"]],
            "loss": scalars[o["loss"]], "I": scalars[o["I"]:o["I"] + D * C_].view(D, C_),
            "card": scalars[o["card"]:o["card"] + D * C_].view(D, C_), "cnt": scalars[o["cnt"]:o["cnt"] + C_]}


class SupOut(NamedTuple):
    loss: torch.Tensor      # sum_k ce_coef*CE_k + dice_coef*Dice_k (differentiable, 0-dim)
    scalars: torch.Tensor


class _SupLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, labels, ce_coef, dice_coef, eps, *logits):
        ctx.set_materialize_grads(False)
        zs, D, B, Cc, H, W = _check_heads(logits, "uaps_sup_loss")
        dev = zs[0].device
        if labels.shape != (B, H, W):
            raise ValueError(f"labels must be [B,H,W]={B, H, W}, got {tuple(labels.shape)}")
        if labels.device != dev:
            raise ValueError("labels and logits on different devices")
        y = labels.to(torch.int64).contiguous()
        L = _lib.lib()
        off = _s_off(D, Cc)
        need = C.c_size_t()
        _lib.check(L.uaps_loss_workspace_bytes(D, B, Cc, H, W, C.byref(need)), "uaps_loss_workspace_bytes")
        ws = _workspace(dev, need.value)
        scalars = torch.empty(off["n"], dtype=torch.float32, device=dev)
        with _lib.device_guard(dev), _timed("uaps_sup_fwd"):
            rc = L.uaps_sup_fwd(_lib.ptr_array(zs), y.data_ptr(), D, B, Cc, H, W, float(ce_coef), float(dice_coef),
                                float(eps), scalars.data_ptr(), ws.data_ptr(), ws.numel(), _lib.current_stream(dev))
        _lib.check(rc, "uaps_sup_fwd")
        ctx.save_for_backward(y, scalars, *zs)
        ctx.meta = (D, B, Cc, H, W, float(ce_coef), float(dice_coef))
        ctx.mark_non_differentiable(scalars)
        return scalars[off["sup"]].clone(), scalars

    @staticmethod
    def backward(ctx, g_loss, _g_scalars):
        y, scalars, *zs = ctx.saved_tensors
        D, B, Cc, H, W, ce_coef, dice_coef = ctx.meta
        if g_loss is None:
            return (None,) * (4 + D)
        dev = zs[0].device
        g = g_loss.contiguous().to(torch.float32)
        dz = [torch.empty_like(z) for z in zs]
        with _lib.device_guard(dev), _timed("uaps_sup_bwd"):
            rc = _lib.lib().uaps_sup_bwd(_lib.ptr_array(zs), y.data_ptr(), scalars.data_ptr(), ce_coef, dice_coef,
                                         g.data_ptr(), D, B, Cc, H, W, _lib.ptr_array(dz), _lib.current_stream(dev))
        _lib.check(rc, "uaps_sup_bwd")
        return (None, None, None, None) + tuple(dz)


def uaps_sup_loss(lab_logits: Sequence[torch.Tensor], labels: torch.Tensor, eps: float = 1e-7) -> SupOut:
    """supervised_loss of UAPS_train.py:194-218: mean over heads of 0.5 (CrossEntropy + dice_loss)."""
    D = len(lab_logits)
    loss, scalars = _SupLoss.apply(labels, 0.5 / D, 0.5 / D, eps, *lab_logits)
    return SupOut(loss, scalars)


def sup_scalars(scalars: torch.Tensor, D: int, C_: int) -> Dict[str, torch.Tensor]:
    o = _s_off(D, C_)
    return {"ce": scalars[o["ce"]:o["ce"] + D], "dice": scalars[o["dice"]:o["dice"] + D], "sup": scalars[o["sup"]],
            "bad_labels": scalars[o["bad"]], "I": scalars[o["I"]:o["I"] + D * C_].view(D, C_),
            "card": scalars[o["card"]:o["card"] + D * C_].view(D, C_), "cnt": scalars[o["cnt"]:o["cnt"] + C_]}


# ---- the reference's own call signatures --------------------------------------------------------

def dice_loss(true: torch.Tensor, logits: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """Drop-in for utilities/pytorch_losses.py:54-89 `dice_loss(true[B,1,H,W], logits[B,C,H,W], eps)`
    (multi-class branch, C >= 2; sums over batch and space, mean over classes)."""
    if true.dim() != 4 or true.shape[1] != 1:
        raise ValueError(f"true must be [B,1,H,W], got {tuple(true.shape)}")
    loss, _ = _SupLoss.apply(true[:, 0], 0.0, 1.0, eps, logits)
    return loss


def ce_loss(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """nn.CrossEntropyLoss()(logits[B,C,H,W], target[B,H,W]) as used at UAPS_train.py:194-197, 259-262."""
    loss, _ = _SupLoss.apply(target, 1.0, 0.0, 1e-7, logits)
    return loss


class _PairLoss(torch.autograd.Function):
    """Supervised branch on rows [:B] and unsupervised branch on rows [B:] of D logit tensors [2B,C,H,W] (the output
    of UNet_UAPS.forward_pair) as ONE forward launch (+ finalize) and ONE backward launch (uaps_pairloss_*); the backward
    writes both halves of the D gradient tensors directly.

    `exchange`: None, or a callable that sums a float64 device tensor of raw loss sums over the ranks IN PLACE and returns the
    number of ranks: the CE means, Dice sums and uncertainty means are then those of the gathered global batch, which is what
    the reference's nn.DataParallel computes (UAPS_model.py:13, UAPS_train.py:194-277); the parameter gradients of the ranks
    must then be ADDED (not averaged), see uaps_amd.dist.GradBuckets(average=False)."""

    @staticmethod
    def forward(ctx, labels, w, cw1, cw2, eps, want_var, exchange, halves, *logits):
        """halves=True: `logits` = D tensors [2B,C,H,W], rows [:B] labelled / rows [B:] unlabelled (forward_pair);
        halves=False: `logits` = D labelled tensors followed by D unlabelled tensors, [B,C,H,W] each."""
        ctx.set_materialize_grads(False)
        if halves:
            zs, D, B2, Cc, H, W = _check_heads(logits, "uaps_pair_loss")
            if B2 % 2:
                raise ValueError("uaps_pair_loss: the batch must hold a labelled and an unlabelled half of equal size")
            B = B2 // 2
            half = B * Cc * H * W * 4
            lab_addr, un_addr = [z.data_ptr() for z in zs], [z.data_ptr() + half for z in zs]
        else:
            if len(logits) % 2:
                raise ValueError("uaps_step_loss: as many labelled as unlabelled heads")
            zl, D, B, Cc, H, W = _check_heads(logits[: len(logits) // 2], "uaps_step_loss")
            zu, D2, Bu, Cu, Hu, Wu = _check_heads(logits[len(logits) // 2:], "uaps_step_loss")
            if (D2, Bu, Cu, Hu, Wu) != (D, B, Cc, H, W):
                raise ValueError("uaps_step_loss: labelled and unlabelled logits must have the same shape for the one-launch form")
            zs = list(zl) + list(zu)
            lab_addr, un_addr = [z.data_ptr() for z in zl], [z.data_ptr() for z in zu]
        dev = zs[0].device
        if labels.shape != (B, H, W) or labels.device != dev:
            raise ValueError(f"labels must be [B,H,W]={B, H, W} on {dev}, got {tuple(labels.shape)} on {labels.device}")
        y = labels.to(torch.int64).contiguous()
        L = _lib.lib()
        lab_p = (C.c_void_p * D)(*lab_addr)
        un_p = (C.c_void_p * D)(*un_addr)
        need = C.c_size_t()
        _lib.check(L.uaps_pairloss_workspace_bytes(D, Cc, C.byref(need)), "uaps_pairloss_workspace_bytes")
        ws = _workspace(dev, need.value)
        so, uo = _s_off(D, Cc), _u_off(D, Cc)
        sscal = torch.empty(so["n"], dtype=torch.float32, device=dev)
        uscal = torch.empty(uo["n"], dtype=torch.float32, device=dev)
        pseudo = torch.empty((B, H, W), dtype=torch.int64, device=dev)
        var = torch.empty((D, B, H, W), dtype=torch.float32, device=dev) if want_var else None
        sums = None
        if exchange is not None:
            ns = C.c_int()
            _lib.check(L.uaps_pairloss_num_sums(D, Cc, C.byref(ns)), "uaps_pairloss_num_sums")
            sums = torch.empty(ns.value, dtype=torch.float64, device=dev)
        w64 = (C.c_double * D)(*[float(x) for x in w]) if w is not None else None      # None: the device step state's weights
        st = _lib.current_stream(dev)
        n_loss = B * H * W
        with _lib.device_guard(dev):
            with _timed("uaps_pair_fwd"):
                rc = L.uaps_pairloss_fwd(lab_p, un_p, y.data_ptr(), w64, D, B, Cc, H, W, float(cw1), float(cw2), float(eps),
                                         pseudo.data_ptr(), var.data_ptr() if want_var else None, sscal.data_ptr(), uscal.data_ptr(),
                                         sums.data_ptr() if sums is not None else None, ws.data_ptr(), ws.numel(), PAIR_CFG, st)
            _lib.check(rc, "uaps_pairloss_fwd")
            if sums is not None:
                n_loss *= int(exchange(sums))                 # the one exchange step of the loss block
                rc = L.uaps_pairloss_finalize_sums(sums.data_ptr(), D, Cc, n_loss, float(cw1), float(cw2), float(eps), sscal.data_ptr(),
                                                   uscal.data_ptr(), st)
                _lib.check(rc, "uaps_pairloss_finalize_sums")
        # the three scalars are aliases of slots the finalize kernel wrote (detach(): plain storage-sharing tensors, not autograd views) --
        # round 5 cloned two of them and added them with an ATen launch: three dependent launches between the loss forward and its backward
        sup, unsup, total = sscal[so["sup"]].detach(), uscal[uo["loss"]].detach(), uscal[uo["total"]].detach()
        ctx.save_for_backward(y, pseudo, sscal, uscal, *zs)
        ctx.meta = (D, B, Cc, H, W, float(cw1), float(cw2), n_loss, bool(halves))
        outs = (total, sup, unsup, pseudo, sscal, uscal) + ((var,) if want_var else ())
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, g_loss, *unused):
        y, pseudo, sscal, uscal, *zs = ctx.saved_tensors
        D, B, Cc, H, W, cw1, cw2, n_loss, halves = ctx.meta
        if g_loss is None:
            return (None,) * (8 + len(zs))
        dev = zs[0].device
        g = g_loss.contiguous().to(torch.float32)
        dz = [torch.empty_like(z) for z in zs]
        L = _lib.lib()
        st = _lib.current_stream(dev)
        if halves:
            half = B * Cc * H * W * 4
            lab_p = (C.c_void_p * D)(*[z.data_ptr() for z in zs])
            un_p = (C.c_void_p * D)(*[z.data_ptr() + half for z in zs])
            dlab_p = (C.c_void_p * D)(*[z.data_ptr() for z in dz])
            dun_p = (C.c_void_p * D)(*[z.data_ptr() + half for z in dz])
        else:
            lab_p = (C.c_void_p * D)(*[z.data_ptr() for z in zs[:D]])
            un_p = (C.c_void_p * D)(*[z.data_ptr() for z in zs[D:]])
            dlab_p = (C.c_void_p * D)(*[z.data_ptr() for z in dz[:D]])
            dun_p = (C.c_void_p * D)(*[z.data_ptr() for z in dz[D:]])
        from . import bounds
        # max|d logits|: operand bound of out_conv's weight gradient -- only where that runs on the matrix core: with <= 4 classes
        # on >= 64-pixel-wide maps it is the exact-N fp32 kernel (csrc/conv_small.hpp), which needs no bound, and the
        # tracking costs the backward kernel 17 us of VALU work (53 -> 70 us at 16 + 16 images of 256 x 256).  On 256-wide maps the
        # bound is wanted again: out_conv's weight gradient then runs the full-width-row kernel (csrc/conv_split_wrw_row.hpp,
        # 4 x ~25 us less than the exact-N kernel at that size)
        row_wrw = W % 256 == 0 and H % 16 == 0
        am = bounds.new_amax(dev) if bounds.enabled() and (row_wrw or not (Cc <= 4 and W >= 64 and W % 4 == 0)) else None
        with _lib.device_guard(dev), _timed("uaps_pair_bwd"):
            rc = L.uaps_pairloss_bwd_h(_lib.mk_hints((), am) if am is not None else None, lab_p, un_p, y.data_ptr(), pseudo.data_ptr(), sscal.data_ptr(), uscal.data_ptr(), cw1, cw2,
                                     g.data_ptr(), D, B, Cc, H, W, n_loss, dlab_p, dun_p, PAIR_CFG, st)
        _lib.check(rc, "uaps_pairloss_bwd")
        for t in dz:
            bounds.put(t, am)
        return (None, None, None, None, None, None, None, None) + tuple(dz)


class StepLoss(NamedTuple):
    loss: torch.Tensor
    sup: torch.Tensor
    unsup: torch.Tensor
    pseudo: torch.Tensor
    var: Optional[torch.Tensor]
    sup_scalars: torch.Tensor
    unsup_scalars: torch.Tensor


def uaps_pair_loss(pair_logits, labels, w, cw1, cw2, eps=1e-7, return_var=False, exchange=None) -> StepLoss:
    """uaps_step_loss for the output of UNet_UAPS.forward_pair: D tensors [2B,C,H,W] whose rows [:B] are the labelled
    batch (supervised branch, UAPS_train.py:194-218) and rows [B:] the unlabelled one (:186-189, 223-282).
    `exchange`: see _PairLoss (gathered-batch loss statistics across ranks)."""
    if w is not None and len(w) != len(pair_logits):
        raise ValueError("one mixing weight per head")
    out = _PairLoss.apply(labels, tuple(float(x) for x in w) if w is not None else None, cw1, cw2, eps, bool(return_var), exchange, True,
                          *pair_logits)
    return StepLoss(out[0], out[1], out[2], out[3], out[6] if return_var else None, out[4], out[5])


def uaps_step_loss(lab_logits, labels, un_logits, w, cw1, cw2, eps=1e-7, return_var=False, exchange=None) -> StepLoss:
    """loss = supervised_loss + cw1*ps_loss + cw2*l_uncert  (UAPS_train.py:282).  With `exchange` (gathered-batch statistics
    across ranks, see _PairLoss) the one-launch pair kernels run on the two logit sets; otherwise the two branch kernels."""
    if exchange is not None:
        if len(w) != len(un_logits):
            raise ValueError("one mixing weight per head")
        out = _PairLoss.apply(labels, tuple(float(x) for x in w), cw1, cw2, eps, bool(return_var), exchange, False,
                              *(tuple(lab_logits) + tuple(un_logits)))
        return StepLoss(out[0], out[1], out[2], out[3], out[6] if return_var else None, out[4], out[5])
    s = uaps_sup_loss(lab_logits, labels, eps)
    u = uaps_unsup_loss(un_logits, w, cw1, cw2, eps, return_var)
    return StepLoss(s.loss + u.loss, s.loss, u.loss, u.pseudo, u.var, s.scalars, u.scalars)
