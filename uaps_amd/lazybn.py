"""BatchNorm(train) + LeakyReLU backward in two halves (include/uaps_hip.h: uaps_bn_act_bwd_prepare / _apply, uaps_call_hints::dyt_*).

The backward of conv -> BatchNorm -> LeakyReLU (UAPS_unet.py:37-43 under autograd) read d(activation) and the raw conv output
twice (the reductions, then dx) and wrote dy for the convolution's two gradient kernels to read again.  Here the node that owns
the BatchNorm runs the reductions only (`prepare`) and hands d(activation) upstream UNTRANSFORMED, registered in `_pending`
under its address; the node that owns the convolution (`take`) lets its weight-gradient kernel form dy while it stages that
operand and write it through for the input-gradient kernel -- or, where the layer's kernel has no such form, runs the
stand-alone pass (`materialize`).  Only raw conv outputs whose producer is one of those nodes are handed up this way (the
producers mark them, `mark`), and `assert_none_pending` after a backward turns a gradient that reached anything else into an
error instead of a silently wrong step.  UAPS_LAZY_BN_BWD=0 keeps the one-piece backward.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

from . import _lib, bounds

_ON = os.environ.get("UAPS_LAZY_BN_BWD", "1") != "0"
_OK = "_uaps_lazy_ok"


class Lazy:
    __slots__ = ("y", "coef", "slope", "groups", "bound")

    def __init__(self, y, coef, slope, groups, bound):
        self.y, self.coef, self.slope, self.groups, self.bound = y, coef, float(slope), int(groups), bound


_pending: Dict[int, Lazy] = {}


def enabled() -> bool:
    return _ON and bounds.enabled()


def mark(y: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """y = conv(., weight) is the raw output of a convolution node that understands a pending transform on y's gradient -- marked
    only where that node's weight-gradient kernel has the in-staging form (csrc/conv_wrw.hip: the full-width-row kernels on
    maps of 256 pixels width or a multiple), because the two-halves backward costs two small launches more than the one-piece one when the stand-alone
    pass has to run after all."""
    Cout, Cin, ks, _ = weight.shape
    if ks == 3 and y.shape[3] % 256 == 0 and y.shape[2] % 16 == 0 and Cout <= 16 and Cin <= 32:
        setattr(y, _OK, True)
    return y


def marked(y: torch.Tensor) -> bool:
    return enabled() and bool(getattr(y, _OK, False))


def prepare(dout, y, gamma, beta, mean, invstd, slope, groups, dgamma, dbeta, dconv_bias, ws) -> Lazy:
    """The reductions of the BatchNorm backward of (dout = d(activation), y); registers dout as pending and returns the record."""
    B, Cc, H, W = y.shape
    dev = y.device
    coef = torch.empty((groups, Cc, 8), dtype=torch.float32, device=dev)
    bnd = bounds.new_amax(dev)
    with _lib.device_guard(dev):
        rc = _lib.lib().uaps_bn_act_bwd_prepare(dout.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                                                invstd.data_ptr(), float(slope), B, Cc, H, W, int(groups), coef.data_ptr(),
                                                dgamma.data_ptr(), dbeta.data_ptr(),
                                                dconv_bias.data_ptr() if dconv_bias is not None else None, bnd.data_ptr(),
                                                ws.data_ptr(), ws.numel(), _lib.current_stream(dev))
    _lib.check(rc, "uaps_bn_act_bwd_prepare")
    lz = Lazy(y, coef, slope, groups, (bnd, 1.0))
    _pending[dout.data_ptr()] = lz
    return lz


def take(dz: Optional[torch.Tensor]) -> Optional[Lazy]:
    """The pending transform of the gradient tensor dz (call before anything that could copy it), or None."""
    if dz is None or not _pending:
        return None
    lz = _pending.pop(dz.data_ptr(), None)
    if lz is not None and (lz.y.shape != dz.shape or lz.y.device != dz.device):      # a stale record of an abandoned backward
        return None
    return lz


def materialize(dz: torch.Tensor, lz: Lazy) -> torch.Tensor:
    """dy by the stand-alone pass; carries the exact max|dy| as its bound."""
    B, Cc, H, W = lz.y.shape
    dz = dz.contiguous()
    dy = torch.empty_like(lz.y)
    am = bounds.new_amax(dz.device) if bounds.enabled() else None
    with _lib.device_guard(dz.device):
        if am is not None:
            _lib.hints((), am)
        rc = _lib.lib().uaps_bn_act_bwd_apply(dz.data_ptr(), lz.y.data_ptr(), lz.coef.data_ptr(), lz.slope, B, Cc, H, W, lz.groups,
                                              dy.data_ptr(), _lib.current_stream(dz.device))
    _lib.check(rc, "uaps_bn_act_bwd_apply")
    return bounds.put(dy, am)


def reset() -> None:
    """Forget records a failed backward may have left behind (call in front of a backward: a stale record could meet a new tensor
    at the same address)."""
    _pending.clear()


def assert_none_pending() -> None:
    if _pending:
        n = len(_pending)
        _pending.clear()
        raise RuntimeError(f"{n} gradient(s) with a pending BatchNorm transform reached a node that does not apply it "
                           "(uaps_amd/lazybn.py); set UAPS_LAZY_BN_BWD=0")


ERANGE = -2
