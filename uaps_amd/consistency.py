"""Sibling consistency / uncertainty terms of the reference on the HIP kernels of csrc/pair_ops.hip.

Same names and call forms as the reference's helper modules, which UAPS_train.py star-imports (lines 21-22):
  softmax_mse_loss, softmax_kl_loss, entropy_map, entropy_minmization   utilities/losses_1.py:9-48, 139-149
  kl_loss                                                                utilities/losses_2.py:201-213
and the test-time uncertainty map of the evaluation notebook (UAPS-Testing.ipynb cell 24):
  uncertainty_map(main_logits, aux_logits) = sum_c KLDivLoss('none')(log_softmax(main), softmax(aux)).
As in the reference, gradients flow to the first argument only.  GPU tensors only.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import torch

from . import _lib

_ws: Dict[Tuple[int, int], torch.Tensor] = {}


def _workspace(dev: torch.device) -> torch.Tensor:
    key = (dev.index, _lib.current_stream(dev))
    w = _ws.get(key)
    if w is None:
        n = C.c_size_t()
        _lib.check(_lib.lib().uaps_pair_workspace_bytes(C.byref(n)), "uaps_pair_workspace_bytes")
        w = _ws[key] = torch.empty(n.value, dtype=torch.uint8, device=dev)
    return w


def _pair(a: torch.Tensor, b: torch.Tensor, what: str):
    _lib.require_device(a, what)
    if a.shape != b.shape or a.dim() != 4:
        raise AssertionError(f"{what}: two [B,C,H,W] tensors of the same size expected")      # losses_1.py:17,37 assert
    if a.dtype != torch.float32 or b.dtype != torch.float32:
        raise TypeError(f"{what}: float32 expected")
    return a.contiguous(), b.contiguous()


def _fwd(a, b, probs, want_mse, want_map, want_mean):
    B, Cc, H, W = a.shape
    dev = a.device
    mse = torch.empty_like(a) if want_mse else None
    klm = torch.empty((B, H, W), dtype=torch.float32, device=dev) if want_map else None
    mean = torch.empty((), dtype=torch.float32, device=dev) if want_mean else None
    ws = _workspace(dev)
    with _lib.device_guard(dev):
        rc = _lib.lib().uaps_softmax_pair_fwd(a.data_ptr(), b.data_ptr(), int(probs), B, Cc, H, W,
                                              mse.data_ptr() if want_mse else None, klm.data_ptr() if want_map else None,
                                              mean.data_ptr() if want_mean else None, ws.data_ptr(), ws.numel(),
                                              _lib.current_stream(dev))
    _lib.check(rc, "uaps_softmax_pair_fwd")
    return mse, klm, mean


class _SoftmaxMse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _pair(a, b, "softmax_mse_loss")
        ctx.save_for_backward(a, b)
        return _fwd(a, b, False, True, False, False)[0]

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da = torch.empty_like(a)
        B, Cc, H, W = a.shape
        with _lib.device_guard(a.device):
            rc = _lib.lib().uaps_softmax_mse_bwd(a.data_ptr(), b.data_ptr(), g.data_ptr(), B, Cc, H, W, da.data_ptr(),
                                                 _lib.current_stream(a.device))
        _lib.check(rc, "uaps_softmax_mse_bwd")
        return da, None


class _SoftmaxKl(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _pair(a, b, "softmax_kl_loss")
        ctx.save_for_backward(a, b)
        return _fwd(a, b, False, False, False, True)[2]

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous().to(torch.float32)
        da = torch.empty_like(a)
        B, Cc, H, W = a.shape
        with _lib.device_guard(a.device):
            rc = _lib.lib().uaps_softmax_kl_bwd(a.data_ptr(), b.data_ptr(), g.data_ptr(), B, Cc, H, W, da.data_ptr(),
                                                _lib.current_stream(a.device))
        _lib.check(rc, "uaps_softmax_kl_bwd")
        return da, None


def softmax_mse_loss(input_logits: torch.Tensor, target_logits: torch.Tensor, sigmoid: bool = False) -> torch.Tensor:
    """utilities/losses_1.py:9-26: (softmax(input) - softmax(target))^2, elementwise map [B,C,H,W]."""
    if sigmoid:
        raise NotImplementedError("the sigmoid variant belongs to the binary heads UAPS never uses")
    return _SoftmaxMse.apply(input_logits, target_logits)


def softmax_kl_loss(input_logits: torch.Tensor, target_logits: torch.Tensor, sigmoid: bool = False) -> torch.Tensor:
    """utilities/losses_1.py:29-48: F.kl_div(log_softmax(input), softmax(target), reduction='mean') (0-dim)."""
    if sigmoid:
        raise NotImplementedError("the sigmoid variant belongs to the binary heads UAPS never uses")
    return _SoftmaxKl.apply(input_logits, target_logits)


def kl_loss(pr: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """utilities/losses_2.py:201-213: F.kl_div(torch.log(pr), gt, reduction='mean') on probabilities (no gradient)."""
    a, b = _pair(pr.detach(), gt.detach(), "kl_loss")
    return _fwd(a, b, True, False, False, True)[2]


def uncertainty_map(main_logits: torch.Tensor, aux_logits: torch.Tensor) -> torch.Tensor:
    """UAPS-Testing.ipynb cell 24: sum over classes of KLDivLoss(reduction='none')(log_softmax(main), softmax(aux)),
    the per-pixel test-time uncertainty map [B,H,W] (no gradient)."""
    a, b = _pair(main_logits.detach(), aux_logits.detach(), "uncertainty_map")
    return _fwd(a, b, False, False, True, False)[1]


def _entropy(p: torch.Tensor, want_map: bool, want_mean: bool):
    _lib.require_device(p, "entropy_map")
    if p.dim() != 4 or p.dtype != torch.float32:
        raise ValueError("entropy_map: float32 [B,C,H,W] probabilities expected")
    p = p.detach().contiguous()
    B, Cc, H, W = p.shape
    dev = p.device
    ent = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev) if want_map else None
    mean = torch.empty((), dtype=torch.float32, device=dev) if want_mean else None
    ws = _workspace(dev)
    with _lib.device_guard(dev):
        rc = _lib.lib().uaps_entropy_map(p.data_ptr(), B, Cc, H, W, ent.data_ptr() if want_map else None,
                                         mean.data_ptr() if want_mean else None, ws.data_ptr(), ws.numel(), _lib.current_stream(dev))
    _lib.check(rc, "uaps_entropy_map")
    return ent, mean


def entropy_map(p: torch.Tensor) -> torch.Tensor:
    """utilities/losses_1.py:146-149: -sum_c p log(p + 1e-6), keepdim -> [B,1,H,W] (no gradient)."""
    return _entropy(p, True, False)[0]


def entropy_minmization(p: torch.Tensor) -> torch.Tensor:
    """utilities/losses_1.py:139-143 (the reference's spelling): mean over pixels of the entropy map (no gradient)."""
    return _entropy(p, False, True)[1]
