"""BatchNorm(train) + LeakyReLU backward in two halves (include/uaps_hip.h: uaps_bn_act_bwd_prepare / _apply, uaps_call_hints::dyt_*).

The backward of conv -> BatchNorm -> LeakyReLU (UAPS_unet.py:37-43 under autograd) read d(activation) and the raw conv output
twice (the reductions, then dx) and wrote dy for the convolution's two gradient kernels to read again.  Here the node that owns
the BatchNorm runs the reductions only (`prepare`) and hands d(activation) upstream UNTRANSFORMED; the node that owns the
convolution (`take`) lets its weight-gradient kernel form dy while it stages that operand and write it through for the
input-gradient kernel -- or, where the layer's kernel has no such form, runs the stand-alone pass (`materialize`).

Safety (round 5).  Between the two nodes the tensor autograd calls "gradient" is not one, so the mechanism is fenced:
  * it is OFF unless the forward runs inside `lazybn.scope()` -- the trainers (UAPSTrainer, BaselineTrainer, StepGraph) open one
    around forward + backward of a step; a user-driven `model(x)` / `loss.backward()` (the reference's
    UAPS_train.py:177-292 loop) always takes the one-piece backward;
  * the pending record travels ON THE GRADIENT TENSOR (as `_uaps_bound` does), together with the tensor's version counter and
    address at hand-over: nothing is keyed by an address, a recycled allocation can never meet a stale record, and a gradient
    that was summed in place with a second consumer's (same object, higher version) is refused with an error instead of being
    transformed; a sum that made a new tensor drops the record and `assert_none_pending` (scope exit) fails the step;
  * a conv output that is observed -- a tensor hook, `retain_grad()` -- is never handed an untransformed gradient (checked when
    the consumer's forward runs and again in its backward): the observer sees the true gradient.
UAPS_LAZY_BN_BWD=0 keeps the one-piece backward everywhere.
"""
from __future__ import annotations

import contextlib
import os
from typing import Optional

import torch

from . import _lib, bounds, config, stepctx

_ON = config.flag("UAPS_LAZY_BN_BWD", True)
_OK = "_uaps_lazy_ok"
_REC = "_uaps_lazy"
_prepared = 0          # records ever handed up (tests)


class Scope:
    """One `with lazybn.scope():` block.  The forward (one thread) asks `current()` for it; the nodes that hand gradients up keep it
    on their ctx, so the count of records handed up and not yet taken belongs to THIS step whichever autograd thread runs its
    backward and whatever other trainers do meanwhile (round 6; the count was a module global)."""
    __slots__ = ("outstanding", "depth")

    def __init__(self):
        self.outstanding = 0
        self.depth = 0


_loose = Scope()       # records prepared by direct calls outside any scope (tests, tools)


def current() -> Optional[Scope]:
    """The scope open on the calling thread, or None."""
    return stepctx.fwd().lazy_scope


class Lazy:
    __slots__ = ("y", "coef", "slope", "groups", "bound", "version", "ptr", "scope")

    def __init__(self, y, coef, slope, groups, bound, version=0, ptr=0, scope=None):
        self.y, self.coef, self.slope, self.groups, self.bound = y, coef, float(slope), int(groups), bound
        self.version, self.ptr, self.scope = version, ptr, scope


@contextlib.contextmanager
def scope():
    """Forward + backward of one training step of a caller that drives both itself (the trainers): inside, marked conv outputs get
    the two-halves backward.  On a clean exit every handed-up gradient must have been taken by its producer; an exception drops
    what a failed backward left behind.  Belongs to the calling thread; nested scopes join the outer one."""
    f = stepctx.fwd()
    sc = f.lazy_scope
    outer = sc is None
    if outer:
        sc = f.lazy_scope = Scope()
    sc.depth += 1
    try:
        yield sc
    except BaseException:
        sc.outstanding = 0
        raise
    finally:
        sc.depth -= 1
        if outer:
            f.lazy_scope = None
    if outer:
        assert_none_pending(sc)


def enabled() -> bool:
    return _ON and stepctx.fwd().lazy_scope is not None and bounds.enabled()


def mark(y: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """y = conv(., weight) is the raw output of a convolution node that understands a pending transform on y's gradient -- marked
    only where that node's weight-gradient kernel has the in-staging form (csrc/conv_wrw.hip: the full-width-row kernels on
    maps of 256 pixels width or a multiple), because the two-halves backward costs two small launches more than the one-piece one when the stand-alone
    pass has to run after all."""
    Cout, Cin, ks, _ = weight.shape
    if ks == 3 and y.shape[3] % 256 == 0 and y.shape[2] % 16 == 0 and Cout <= 16 and Cin <= 32:
        setattr(y, _OK, True)
    return y


def observed(y: torch.Tensor) -> bool:
    """Someone other than y's producer will look at y's gradient: a tensor hook or retain_grad()."""
    return bool(getattr(y, "_backward_hooks", None)) or (y.requires_grad and not y.is_leaf and y.retains_grad)


def marked(y: torch.Tensor) -> bool:
    return enabled() and bool(getattr(y, _OK, False)) and not observed(y)


def prepare(dout, y, gamma, beta, mean, invstd, slope, groups, dgamma, dbeta, dconv_bias, ws, scope: Optional[Scope] = None) -> Lazy:
    """The reductions of the BatchNorm backward of (dout = d(activation), y); attaches the pending record to dout and returns it.
    scope: the Scope the owning node's forward ran in (kept on its ctx), else the loose count."""
    global _prepared
    scope = scope or _loose
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
    lz = Lazy(y, coef, slope, groups, (bnd, 1.0), dout._version, dout.data_ptr(), scope)
    setattr(dout, _REC, lz)
    scope.outstanding += 1
    _prepared += 1
    return lz


def prepare_from_partials(dout, y, gamma, beta, mean, invstd, slope, groups, dgamma, dbeta, dconv_bias, partials, maxes,
                          scope: Optional[Scope] = None) -> Lazy:
    """`prepare` whose reductions were formed in the epilogue of the kernel that wrote dout (conv.conv_bwd_data_raw bsum:
    partials float2 [C][B][parts], maxes = max|d| and max|x_hat|): the finalize alone."""
    global _prepared
    scope = scope or _loose
    B, Cc, H, W = y.shape
    dev = y.device
    coef = torch.empty((groups, Cc, 8), dtype=torch.float32, device=dev)
    bnd = bounds.new_amax(dev)
    with _lib.device_guard(dev):
        rc = _lib.lib().uaps_bn_act_bwd_finalize(partials.data_ptr(), int(partials.shape[2]), maxes.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                 mean.data_ptr(), invstd.data_ptr(), B, Cc, H, W, int(groups), coef.data_ptr(), dgamma.data_ptr(),
                                                 dbeta.data_ptr(), dconv_bias.data_ptr() if dconv_bias is not None else None, bnd.data_ptr(),
                                                 _lib.current_stream(dev))
    _lib.check(rc, "uaps_bn_act_bwd_finalize")
    lz = Lazy(y, coef, slope, groups, (bnd, 1.0), dout._version, dout.data_ptr(), scope)
    setattr(dout, _REC, lz)
    scope.outstanding += 1
    _prepared += 1
    return lz


def prepared_total() -> int:
    return _prepared


def take(dz: Optional[torch.Tensor]) -> Optional[Lazy]:
    """The pending transform of the gradient tensor dz (call before anything that could copy it), or None.  Raises when dz is
    no longer the tensor that was handed up (summed in place with another consumer's gradient, resized, re-pointed)."""
    if dz is None:
        return None
    lz = getattr(dz, _REC, None)
    if lz is None:
        return None
    delattr(dz, _REC)
    if lz.scope is not None:
        lz.scope.outstanding = max(0, lz.scope.outstanding - 1)
    if dz._version != lz.version or dz.data_ptr() != lz.ptr or dz.shape != lz.y.shape or dz.device != lz.y.device:
        raise RuntimeError("a gradient with a pending BatchNorm transform was modified before its producer saw it (a second "
                           "consumer of a raw conv output inside lazybn.scope()?); set UAPS_LAZY_BN_BWD=0")
    return lz


def materialize(dz: torch.Tensor, lz: Lazy) -> torch.Tensor:
    """dy by the stand-alone pass; carries the exact max|dy| as its bound."""
    B, Cc, H, W = lz.y.shape
    dz = dz.contiguous()
    dy = torch.empty_like(lz.y)
    am = bounds.new_amax(dz.device) if bounds.enabled() else None
    with _lib.device_guard(dz.device):
        rc = _lib.lib().uaps_bn_act_bwd_apply_h(_lib.mk_hints((), am) if am is not None else None, dz.data_ptr(), lz.y.data_ptr(), lz.coef.data_ptr(), lz.slope, B, Cc, H, W, lz.groups,
                                              dy.data_ptr(), _lib.current_stream(dz.device))
    _lib.check(rc, "uaps_bn_act_bwd_apply")
    return bounds.put(dy, am)


def reset() -> None:
    """Forget what a failed backward may have left behind (records live on their gradient tensors and die with them; only the
    counts remain: the calling thread's open scope and the loose one)."""
    _loose.outstanding = 0
    sc = current()
    if sc is not None:
        sc.outstanding = 0


def assert_none_pending(sc: Optional[Scope] = None) -> None:
    """Raises when `sc` (default: the calling thread's open scope, else the loose count) still has records nobody took."""
    sc = sc or current() or _loose
    if sc.outstanding:
        n, sc.outstanding = sc.outstanding, 0
        raise RuntimeError(f"{n} gradient(s) with a pending BatchNorm transform reached a node that does not apply it "
                           "(uaps_amd/lazybn.py); set UAPS_LAZY_BN_BWD=0")


ENOFORM = -4        # UAPS_ENOFORM (include/uaps_hip.h): the layer's kernel cannot apply the pending transform while staging
