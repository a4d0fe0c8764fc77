"""Host side of the ConvBlock / UpBlock glue kernels (csrc/norm_act.hip, csrc/resample.hip):
fused BatchNorm(train) + LeakyReLU + Dropout and bilinear-x2 + channel concat, with their backwards.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _graddest, lazybn, _lib, bounds, stepctx
from . import perturb as _perturb

_ws: Dict[Tuple[int, int], torch.Tensor] = {}

# Number of statistics groups of the train-mode BatchNorms: the batch is `_groups()` consecutive blocks, each
# normalised on its own (UNet_UAPS.forward_pair runs the labelled and the unlabelled batch of a step as one
# 2-group batch, which is what the reference's two separate forwards compute, UAPS_train.py:177,185).
def _groups() -> int:
    return stepctx.fwd().stat_groups


class stat_groups:
    """Context manager: `with stat_groups(2): model(x)`.  Per thread (a forward runs on one thread)."""

    def __init__(self, n: int):
        self.n = int(n)

    def __enter__(self):
        f = stepctx.fwd()
        self.prev, f.stat_groups = f.stat_groups, self.n
        return self

    def __exit__(self, *a):
        stepctx.fwd().stat_groups = self.prev
        return False


def _workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    key = (dev.index, _lib.current_stream(dev))
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _ws[key] = w
    return w


def _bn_ws(dev, B, Cc, H, W):
    n = C.c_size_t()
    _lib.check(_lib.lib().uaps_bn_workspace_bytes(B, Cc, H, W, C.byref(n)), "uaps_bn_workspace_bytes")
    return _workspace(dev, n.value)


class _BnActTrain(torch.autograd.Function):
    """y (conv output, no bias) -> dropout(leaky_relu(batch_norm_train(y + conv_bias)))."""

    @staticmethod
    def forward(ctx, y, conv_bias, gamma, beta, running_mean, running_var, nbt, momentum, eps, slope, drop_p, seed, offset,
                groups=1, stats_partials=None):
        _lib.require_device(y, "bn_act")
        y = y.contiguous()
        B, Cc, H, W = y.shape
        if B % groups:
            raise ValueError(f"bn_act: batch {B} is not divisible into {groups} statistics groups")
        dev = y.device
        out = torch.empty_like(y)
        stats = torch.empty((2, groups * Cc), dtype=torch.float32, device=dev)
        ws = _bn_ws(dev, B, Cc, H, W)
        if stats_partials is not None:
            if stats_partials.shape[:2] != (Cc, B) or stats_partials.shape[-1] != 2 or not stats_partials.is_contiguous():
                raise ValueError("bn_act: stats must be the [C, B, parts, 2] tensor conv2d_with_stats returned for this y")
            fn, head = _lib.lib().uaps_bn_act_fwd_train_partials_h, (stats_partials.data_ptr(), int(stats_partials.shape[2]))
        else:
            fn, head = _lib.lib().uaps_bn_act_fwd_train_grouped, ()
        with _lib.device_guard(dev):
            if stats_partials is not None:       # hints: the shift the producing conv formed its sums about
                head = (_lib.mk_hints((), None, (running_mean, conv_bias)) if getattr(stats_partials, "_uaps_shifted", False) else None,) + head
            rc = fn(
                *head, y.data_ptr(), conv_bias.data_ptr() if conv_bias is not None else None, gamma.data_ptr(), beta.data_ptr(),
                running_mean.data_ptr() if running_mean is not None else None,
                running_var.data_ptr() if running_var is not None else None,
                nbt.data_ptr() if nbt is not None else None, float(momentum), float(eps), float(slope), float(drop_p),
                seed, offset, B, Cc, H, W, groups, out.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), ws.data_ptr(),
                ws.numel(), _lib.current_stream(dev))
        _lib.check(rc, "uaps_bn_act_fwd_train_grouped")
        ctx.save_for_backward(y, gamma, beta, stats)
        ctx.meta = (float(slope), float(drop_p), seed, offset, conv_bias is not None, groups)
        ctx.keys = (id(gamma), id(beta), id(conv_bias) if conv_bias is not None else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, gamma, beta, stats = ctx.saved_tensors
        slope, drop_p, seed, offset, has_bias, groups = ctx.meta
        dout = dout.contiguous()
        B, Cc, H, W = y.shape
        dev = y.device
        dy = torch.empty_like(y)
        # dgamma, dbeta, d(conv bias); the conv bias feeds a train-mode BatchNorm, so its gradient (the sum of dy
        # over a channel) is exactly zero -- written by the same finalize kernel instead of a fill launch
        dgb = [_graddest.take(k, (Cc,), dev) for k in ctx.keys]
        ws = _bn_ws(dev, B, Cc, H, W)
        am = bounds.new_amax(dev) if bounds.enabled() else None      # max|dy|: the operand bound of the convolution's backward
        with _lib.device_guard(dev):
            rc = _lib.lib().uaps_bn_act_bwd_grouped_bias_h(_lib.mk_hints((), am) if am is not None else None, dout.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                         stats[0].data_ptr(), stats[1].data_ptr(), slope, drop_p, seed, offset, B,
                                                         Cc, H, W, groups, dy.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(),
                                                         dgb[2].data_ptr(), ws.data_ptr(), ws.numel(), _lib.current_stream(dev))
        _lib.check(rc, "uaps_bn_act_bwd_grouped_bias")
        bounds.put(dy, am)
        return dy, (dgb[2] if has_bias else None), dgb[0], dgb[1], None, None, None, None, None, None, None, None, None, None, None


class _BnAddRelu(torch.autograd.Function):
    """relu(batch_norm_train(y) + identity) -- the end of a residual block (utilities/resnet.py:44-50, 85-91) -- in the BatchNorm's
    apply pass, handed out as `n` handles on one tensor (one per consumer, see res_uaps.add_relu).  The backward sums the consumers'
    gradients behind the ReLU mask in one pass (uaps_relu_bwd_sum); that sum is the identity's gradient and, through the usual
    BatchNorm backward with an identity activation, y's."""

    @staticmethod
    def forward(ctx, y, identity, gamma, beta, running_mean, running_var, nbt, momentum, eps, groups, stats_partials, n, am):
        _lib.require_device(y, "bn_add_relu")
        ctx.set_materialize_grads(False)
        y, identity = y.contiguous(), identity.contiguous()
        if identity.data_ptr() % 16:              # a contiguous view at an odd storage offset: the apply pass streams 16-byte pieces of it
            identity = identity.clone()
        B, Cc, H, W = y.shape
        if identity.shape != y.shape or B % groups or identity.device != y.device:
            raise ValueError(f"bn_add_relu: y {tuple(y.shape)}, identity {tuple(identity.shape)} on {identity.device}, {groups} statistics groups")
        if stats_partials.shape[:2] != (Cc, B) or stats_partials.shape[-1] != 2 or not stats_partials.is_contiguous():
            raise ValueError("bn_add_relu: stats must be the [C, B, parts, 2] tensor conv2d_with_stats returned for this y")
        dev = y.device
        out = torch.empty_like(y)
        stats = torch.empty((2, groups * Cc), dtype=torch.float32, device=dev)
        ws = _bn_ws(dev, B, Cc, H, W)
        with _lib.device_guard(dev):
            shifted = getattr(stats_partials, "_uaps_shifted", False)
            rc = _lib.lib().uaps_bn_act_fwd_train_partials_h(
                _lib.mk_hints((), am, (running_mean, None) if shifted else None, residual=identity), stats_partials.data_ptr(), int(stats_partials.shape[2]), y.data_ptr(), None, gamma.data_ptr(), beta.data_ptr(),
                running_mean.data_ptr() if running_mean is not None else None,
                running_var.data_ptr() if running_var is not None else None, nbt.data_ptr() if nbt is not None else None,
                float(momentum), float(eps), 1.0, 0.0, 0, 0, B, Cc, H, W, groups, out.data_ptr(), stats[0].data_ptr(),
                stats[1].data_ptr(), ws.data_ptr(), ws.numel(), _lib.current_stream(dev))
        _lib.check(rc, "uaps_bn_act_fwd_train_partials")
        ctx.save_for_backward(y, gamma, beta, stats, out)
        ctx.groups = groups
        ctx.keys = (id(gamma), id(beta))
        return tuple(out.view_as(out) for _ in range(n))

    @staticmethod
    def backward(ctx, *douts):
        y, gamma, beta, stats, out = ctx.saved_tensors
        gs = [g.contiguous() for g in douts if g is not None]
        if not gs:
            return (None,) * 13
        B, Cc, H, W = y.shape
        dev = y.device
        d = torch.empty_like(y)
        dy = torch.empty_like(y)
        dgb = [_graddest.take(k, (Cc,), dev) for k in ctx.keys]
        ws = _bn_ws(dev, B, Cc, H, W)
        am = bounds.new_amax(dev) if bounds.enabled() else None
        L = _lib.lib()
        with _lib.device_guard(dev):
            st = _lib.current_stream(dev)
            arr = (C.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
            _lib.check(L.uaps_relu_bwd_sum(arr, len(gs), out.data_ptr(), d.data_ptr(), out.numel(), st), "uaps_relu_bwd_sum")
            rc = L.uaps_bn_act_bwd_grouped_h(_lib.mk_hints((), am) if am is not None else None, d.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), stats[0].data_ptr(),
                                           stats[1].data_ptr(), 1.0, 0.0, 0, 0, B, Cc, H, W, ctx.groups, dy.data_ptr(), dgb[0].data_ptr(),
                                           dgb[1].data_ptr(), ws.data_ptr(), ws.numel(), st)
        _lib.check(rc, "uaps_bn_act_bwd_grouped")
        bounds.put(dy, am)
        return dy, d, dgb[0], dgb[1], None, None, None, None, None, None, None, None, None


def bn_add_relu(y: torch.Tensor, stats: torch.Tensor, bn: nn.BatchNorm2d, identity: torch.Tensor, n: int = 1):
    """relu(bn_train(y) + identity) with `y`, `stats` from conv2d_with_stats; n > 1: a tuple of n handles on the result."""
    if not 1 <= n <= 4:
        raise ValueError("bn_add_relu: 1..4 consumers")
    mom = 0.1 if bn.momentum is None else bn.momentum
    am = bounds.new_amax(y.device) if bounds.enabled() else None      # max(out), raised by the apply pass: the join feeds the next block's convolutions
    outs = tuple(bounds.put(o, am) for o in _BnAddRelu.apply(y, identity, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                                            bn.num_batches_tracked, mom, bn.eps, _groups(), stats, n, am))
    return outs[0] if n == 1 else outs


class _BnActEval(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, conv_bias, gamma, beta, running_mean, running_var, eps, slope):
        _lib.require_device(y, "bn_act")
        y = y.contiguous()
        B, Cc, H, W = y.shape
        dev = y.device
        out = torch.empty_like(y)
        mean_eff = torch.empty(Cc, dtype=torch.float32, device=dev)
        ws = _bn_ws(dev, B, Cc, H, W)
        with _lib.device_guard(dev):
            rc = _lib.lib().uaps_bn_act_fwd_eval(y.data_ptr(), conv_bias.data_ptr() if conv_bias is not None else None,
                                                 gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(),
                                                 running_var.data_ptr(), float(eps), float(slope), B, Cc, H, W,
                                                 out.data_ptr(), mean_eff.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 _lib.current_stream(dev))
        _lib.check(rc, "uaps_bn_act_fwd_eval")
        ctx.save_for_backward(y, gamma, beta, mean_eff, running_var)
        ctx.meta = (float(eps), float(slope))
        return out

    @staticmethod
    def backward(ctx, dout):
        y, gamma, beta, mean_eff, rv = ctx.saved_tensors
        eps, slope = ctx.meta
        dout = dout.contiguous()
        B, Cc, H, W = y.shape
        dy = torch.empty_like(y)
        ws = _bn_ws(y.device, B, Cc, H, W)
        with _lib.device_guard(y.device):
            rc = _lib.lib().uaps_bn_act_bwd_eval(dout.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                 mean_eff.data_ptr(), rv.data_ptr(), eps, slope, B, Cc, H, W, dy.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), _lib.current_stream(y.device))
        _lib.check(rc, "uaps_bn_act_bwd_eval")
        # parameters are not trained through an eval() forward in the reference; only the input gradient is provided
        return dy, None, None, None, None, None, None, None


def bn_act(y: torch.Tensor, conv_bias: Optional[torch.Tensor], bn: nn.BatchNorm2d, slope: float, drop_p: float,
           training: bool, stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dropout_p(leaky_relu(bn(y + conv_bias))) with nn.BatchNorm2d / nn.LeakyReLU / nn.Dropout semantics
    (UAPS_unet.py:38-40), as three streaming kernels."""
    if training or not bn.track_running_stats:
        p = float(drop_p) if training else 0.0
        seed, off = _perturb.rng().reserve(y.numel()) if p > 0 else (0, 0)
        mom = 0.1 if bn.momentum is None else bn.momentum
        out = _BnActTrain.apply(y, conv_bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                mom, bn.eps, slope, p, seed, off, _groups(), stats)
        if y.is_cuda and slope <= 1.0:             # |gamma x_hat + beta| <= sqrt(n) max(|gamma| + |beta|), dropout scales by 1 / (1 - p)
            b = bounds.bn_output_bound(bn, y.shape[0] // _groups() * y.shape[2] * y.shape[3], 1.0 / (1.0 - p))
            if b is not None:
                bounds.put(out, *b)
        return out
    return _BnActEval.apply(y, conv_bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, slope)


class _BnActConv(torch.autograd.Function):
    """y (raw output of a conv, no bias) -> conv2(leaky_relu(batch_norm_train(y + conv_bias))): the activated tensor
    between the two convs (UAPS_unet.py:38-41) is never written.  The statistics come from the first conv's epilogue
    partials; the second conv normalises + activates while it stages its input (forward and weight gradient), and the
    backward is conv2's input gradient followed by the usual BatchNorm backward on (that gradient, y)."""

    @staticmethod
    def forward(ctx, y, stats_partials, conv_bias, gamma, beta, running_mean, running_var, nbt, momentum, eps, slope, groups,
                weight, bias, want_stats, xb=None, stat_shift=None):
        from . import conv as _conv
        _lib.require_device(y, "bn_act_conv")
        ctx.set_materialize_grads(False)
        y = y.contiguous()
        B, Cc, H, W = y.shape
        Cout, Cin, ks, _ = weight.shape
        if Cin != Cc:
            raise ValueError(f"bn_act_conv: weight expects {Cin} input channels, y has {Cc}")
        if B % groups:
            raise ValueError(f"bn_act_conv: batch {B} is not divisible into {groups} statistics groups")
        if stats_partials.shape[:2] != (Cc, B) or stats_partials.shape[-1] != 2 or not stats_partials.is_contiguous():
            raise ValueError("bn_act_conv: stats must be the [C, B, parts, 2] tensor conv2d_with_stats returned for this y")
        dev = y.device
        L = _lib.lib()
        stats = torch.empty((2, groups * Cc), dtype=torch.float32, device=dev)
        xf = torch.empty((groups, Cc, 2), dtype=torch.float32, device=dev)
        wf, wb = _conv.pack_weights(weight, need_bwd=True)
        z = torch.empty((B, Cout, H, W), dtype=torch.float32, device=dev)
        cfg = _conv.plan_cfg(ks, 0, False)          # a 1x1 here: never the GEMM-tiled plan (conv.plan_cfg)
        fcfg = cfg      # (no UAPS_CONV_BOUNDED here: the staging-time BatchNorm form keeps the tile kernels and their partial-sum layout)
        zstats = None
        if want_stats:
            zstats = torch.empty((Cout, B, _conv.stats_parts_per_image(B, Cin, Cout, H, W, ks, fcfg), 2), dtype=torch.float32, device=dev)
        with _lib.device_guard(dev):
            st = _lib.current_stream(dev)
            rc = L.uaps_bn_finalize_train_h(_lib.mk_hints((), None, (running_mean, conv_bias)) if getattr(stats_partials, "_uaps_shifted", False) else None,
                                            stats_partials.data_ptr(), int(stats_partials.shape[2]),
                                          conv_bias.data_ptr() if conv_bias is not None else None, gamma.data_ptr(),
                                          beta.data_ptr(), running_mean.data_ptr() if running_mean is not None else None,
                                          running_var.data_ptr() if running_var is not None else None,
                                          nbt.data_ptr() if nbt is not None else None, float(momentum), float(eps), B, Cc, H, W,
                                          groups, stats[0].data_ptr(), stats[1].data_ptr(), xf.data_ptr(), st)
            _lib.check(rc, "uaps_bn_finalize_train")
            am = _conv._claim_amax(dev)               # conv.request_out_amax(): track max|z| (the 1x1 projection in front of an up-sampling)
            for attempt in range(2):
                with _conv._timed("fwd_bn", B, Cin, Cout, H, W, ks, fcfg, _conv._h16(xb), want_stats) as tm:
                    hh = None
                    if xb is not None or (want_stats and stat_shift is not None) or am is not None:
                        hh = _lib.mk_hints((xb,) if xb is not None else (), am, stat_shift if want_stats else None)
                    rc = L.uaps_conv_fwd_bn_h(hh, y.data_ptr(), xf.data_ptr(), float(slope), groups, wf.data_ptr(),
                                            bias.data_ptr() if bias is not None else None, z.data_ptr(),
                                            zstats.data_ptr() if want_stats else None, B, Cin, Cout, H, W, ks, fcfg, st)
                    if rc == _conv.ENOFORM:
                        tm.on = False
                if rc != _conv.ENOFORM or am is None:
                    break
                am = None                             # this layer's kernel cannot track it: run without (the caller falls back)
            _lib.check(rc, "uaps_conv_fwd_bn")
            _conv._set_out_amax(am)
        ctx.save_for_backward(y, gamma, beta, stats, xf, wb)
        ctx.meta = (float(slope), groups, conv_bias is not None, bias is not None, Cout, ks, cfg)
        ctx.keys = (id(gamma), id(beta), id(conv_bias) if conv_bias is not None else None, id(weight), id(bias) if bias is not None else None)
        ctx.prefs = _conv.leaf_refs(weight, bias)
        ctx.step = _conv.step_of(ctx.prefs)
        ctx.xb = xb
        ctx.lazy_scope = lazybn.current()
        ctx.lazy_up = lazybn.marked(y) and y.requires_grad      # y's producer applies a pending BatchNorm transform (and will run): the backward hands d(activation) up
        if want_stats:
            zstats._uaps_shifted = stat_shift is not None
            ctx.mark_non_differentiable(zstats)
            return z, zstats
        return z

    @staticmethod
    def backward(ctx, dz, *_unused):
        from . import conv as _conv
        if dz is None:
            return (None,) * 17
        y, gamma, beta, stats, xf, wb = ctx.saved_tensors
        slope, groups, has_cbias, has_bias, Cout, ks, cfg = ctx.meta
        lz_in = lazybn.take(dz)                 # dz is d(activation) behind the BatchNorm that follows conv2: its weight gradient runs first
        dzb, xb = bounds.get(dz), ctx.xb
        dz = dz.contiguous()
        B, Cc, H, W = y.shape
        dev = y.device
        L = _lib.lib()
        n = C.c_size_t()
        _lib.check(L.uaps_conv_wrw_workspace_bytes(B, Cc, Cout, H, W, ks, cfg, C.byref(n)), "uaps_conv_wrw_workspace_bytes")
        cws = _conv._wrw_workspace(dev, n.value, ctx.prefs, ctx.step)
        dw = _graddest.take(ctx.keys[3], (Cout, Cc, ks, ks), dev)
        want_db = has_bias and ctx.needs_input_grad[13]
        db = _graddest.take(ctx.keys[4], (Cout,), dev) if want_db else None

        def weight_gradient(dz, dzb, lz):
            """conv2's weight gradient; with a pending transform the kernel writes the true dz through (None: no such form here)"""
            out = torch.empty_like(dz) if lz is not None else None
            with _lib.device_guard(dev):
                st = _lib.current_stream(dev)
                with _conv._timed("wrw_bn", B, Cc, Cout, H, W, ks, cfg, _conv._h16(dzb, xb), dt=lz is not None) as tm:
                    hh = None
                    if lz is not None:
                        hh = _lib.mk_hints((dzb, xb), dyt=(lz.y, lz.coef, out, lz.slope, lz.groups))
                    elif dzb is not None and xb is not None:
                        hh = _lib.mk_hints((dzb, xb))
                    rc = L.uaps_conv_bwd_weight_partial_bn_h(hh, dz.data_ptr(), y.data_ptr(), xf.data_ptr(), slope, groups, int(want_db), B, Cc,
                                                           Cout, H, W, ks, cfg, cws.data_ptr(), cws.numel(), st)
                    if lz is not None and rc == lazybn.ENOFORM:
                        tm.on = False
                if lz is not None and rc == lazybn.ENOFORM:
                    return None
                _lib.check(rc, "uaps_conv_bwd_weight_partial_bn")
                _conv._wrw_reduce(cws, dw, db, B, Cc, Cout, H, W, ks, cfg, st, ctx.prefs, ctx.step)
            return bounds.put(out, *lz.bound) if lz is not None else dz

        done_w = False
        if lz_in is not None:
            out = weight_gradient(dz, lz_in.bound, lz_in) if xb is not None else None
            done_w = out is not None
            dz = out if done_w else lazybn.materialize(dz, lz_in)
            dzb = bounds.get(dz)
        lazy_up = ctx.lazy_up and not lazybn.observed(y)      # (a hook / retain_grad registered on y since the forward gets the true gradient)
        partials = maxes = None
        if lazy_up:      # where the input-gradient kernel can, its epilogue forms this BatchNorm's backward sums (no pass of their own)
            da, partials, maxes = _conv.conv_bwd_data_raw(dz, wb, Cc, ks, cfg, dyb=dzb, bsum=(y, stats[0], stats[1], gamma, beta, slope, groups))
        else:
            da = _conv.conv_bwd_data_raw(dz, wb, Cc, ks, cfg, dyb=dzb)
        if not done_w:
            weight_gradient(dz, dzb, None)
        dgb = [_graddest.take(k, (Cc,), dev) for k in ctx.keys[:3]]      # dgamma, dbeta, d(conv bias) = 0
        ws = _bn_ws(dev, B, Cc, H, W)
        if lazy_up:
            # the reductions only; d(activation) goes up as it is, its transform pending (lazybn): y's producer forms dy in its weight gradient
            if partials is not None:
                lazybn.prepare_from_partials(da, y, gamma, beta, stats[0], stats[1], slope, groups, dgb[0], dgb[1], dgb[2], partials, maxes, scope=ctx.lazy_scope)
            else:
                lazybn.prepare(da, y, gamma, beta, stats[0], stats[1], slope, groups, dgb[0], dgb[1], dgb[2], ws, scope=ctx.lazy_scope)
            dy = da
        else:
            dy = torch.empty_like(y)
            with _lib.device_guard(dev):
                st = _lib.current_stream(dev)
                am = bounds.new_amax(dev) if bounds.enabled() else None
                rc = L.uaps_bn_act_bwd_grouped_bias_h(_lib.mk_hints((), am) if am is not None else None, da.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), stats[0].data_ptr(),
                                                    stats[1].data_ptr(), slope, 0.0, 0, 0, B, Cc, H, W, groups, dy.data_ptr(),
                                                    dgb[0].data_ptr(), dgb[1].data_ptr(), dgb[2].data_ptr(), ws.data_ptr(), ws.numel(), st)
                _lib.check(rc, "uaps_bn_act_bwd_grouped_bias")
            bounds.put(dy, am)
        return dy, None, (dgb[2] if has_cbias else None), dgb[0], dgb[1], None, None, None, None, None, None, None, dw, db, None, None, None


def can_fuse_bn_into_conv(y: torch.Tensor, weight: torch.Tensor) -> bool:
    """The staging-time BatchNorm of bn_act_conv needs 16-byte rows and the 8-channel-chunk kernels."""
    return y.is_cuda and y.shape[3] % 4 == 0 and y.shape[1] > 4 and weight.shape[2] in (1, 3) and _groups() <= 8


def bn_act_conv(y: torch.Tensor, stats: torch.Tensor, conv_bias: Optional[torch.Tensor], bn: nn.BatchNorm2d, slope: float,
                weight: torch.Tensor, bias: Optional[torch.Tensor], want_stats: bool = False, stat_shift=None):
    """conv2d(leaky_relu(bn_train(y + conv_bias)), weight, bias) (+ the epilogue statistics of the result) where `y`,
    `stats` come from conv2d_with_stats: train-mode only, no dropout between the two (decoder ConvBlocks)."""
    mom = 0.1 if bn.momentum is None else bn.momentum
    xb = bounds.bn_output_bound(bn, y.shape[0] // _groups() * y.shape[2] * y.shape[3]) if slope <= 1.0 else None
    res = _BnActConv.apply(y, stats, conv_bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                           mom, bn.eps, slope, _groups(), weight, bias, want_stats, xb, stat_shift)
    if want_stats:
        res[1]._uaps_shifted = stat_shift is not None
        lazybn.mark(res[0], weight)        # z feeds a BatchNorm of its own: its gradient may arrive with that transform pending
    return res


class _UpCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, skip, low):
        _lib.require_device(low, "up_cat")
        skip, low = skip.contiguous(), low.contiguous()
        B, Cl, h, w = low.shape
        Cs = skip.shape[1]
        if skip.shape != (B, Cs, 2 * h, 2 * w):
            raise ValueError(f"skip {tuple(skip.shape)} is not twice the size of low {tuple(low.shape)}")
        out = torch.empty((B, Cs + Cl, 2 * h, 2 * w), dtype=torch.float32, device=low.device)
        with _lib.device_guard(low.device):
            rc = _lib.lib().uaps_up_cat_fwd(skip.data_ptr(), low.data_ptr(), out.data_ptr(), B, Cs, Cl, h, w,
                                            _lib.current_stream(low.device))
        _lib.check(rc, "uaps_up_cat_fwd")
        ctx.meta = (B, Cs, Cl, h, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, Cs, Cl, h, w = ctx.meta
        dout = dout.contiguous()
        need_skip = ctx.needs_input_grad[0]
        dskip = torch.empty((B, Cs, 2 * h, 2 * w), dtype=torch.float32, device=dout.device) if need_skip else None
        dlow = torch.empty((B, Cl, h, w), dtype=torch.float32, device=dout.device)
        with _lib.device_guard(dout.device):
            rc = _lib.lib().uaps_up_cat_bwd(dout.data_ptr(), dskip.data_ptr() if need_skip else None, dlow.data_ptr(), B, Cs,
                                            Cl, h, w, _lib.current_stream(dout.device))
        _lib.check(rc, "uaps_up_cat_bwd")
        return dskip, dlow


def up_cat(skip: torch.Tensor, low: torch.Tensor) -> torch.Tensor:
    """torch.cat([skip, Upsample(x2, bilinear, align_corners=True)(low)], dim=1)  (UAPS_unet.py:83-85)."""
    return _UpCat.apply(skip, low)


class _FanOut(torch.autograd.Function):
    """x -> n aliases of x; the backward sums the n incoming gradients with ONE kernel (in[0] + in[1] + ... left to
    right) instead of autograd's n-1 separate accumulation adds."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g.contiguous() for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        out = torch.empty_like(gs[0])
        acc = gs
        while len(acc) > 1:                     # the kernel takes up to 4 operands
            chunk, acc = acc[:4], acc[4:]
            ptrs = (C.c_void_p * len(chunk))(*[g.data_ptr() for g in chunk])
            with _lib.device_guard(out.device):
                rc = _lib.lib().uaps_sum_tensors(ptrs, len(chunk), out.data_ptr(), out.numel(), _lib.current_stream(out.device))
            _lib.check(rc, "uaps_sum_tensors")
            acc = [out] + acc
        return out, None


def fan_out(x: torch.Tensor, n: int):
    """n handles on the same tensor whose gradients are summed by one kernel (GPU tensors only)."""
    _lib.require_device(x, "fan_out")
    return tuple(bounds.carry(x, h) for h in _FanOut.apply(x, n))


def cat_batches(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.cat([a, b], dim=0) of the step's two input batches (forward_pair) by one kernel that also takes max|.| of the result:
    the bound lets the first convolution's weight gradient run in the fp16 form (csrc/conv_split_wrw_row.hpp).  Inputs that
    take part in autograd, or are not fp32 on the GPU, go through torch.cat."""
    if not (a.is_cuda and b.is_cuda and a.device == b.device and a.dtype == b.dtype == torch.float32 and a.shape == b.shape) \
            or a.requires_grad or b.requires_grad:
        return torch.cat([a, b], dim=0)           # (two devices: torch.cat raises, as it should -- the kernel would read a foreign pointer)
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty((2 * a.shape[0],) + tuple(a.shape[1:]), dtype=torch.float32, device=a.device)
    am = bounds.new_amax(a.device) if bounds.enabled() else None
    with _lib.device_guard(a.device):
        rc = _lib.lib().uaps_cat2_h(_lib.mk_hints((), am) if am is not None else None, a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _lib.current_stream(a.device))
    _lib.check(rc, "uaps_cat2")
    return bounds.put(out, am) if am is not None else out




class _Up2x(torch.autograd.Function):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (UAPS_unet.py:74-75) on its own: the kernels
    of up_cat with an empty skip part."""

    @staticmethod
    def forward(ctx, low):
        _lib.require_device(low, "upsample2x")
        low = low.contiguous()
        B, Cl, h, w = low.shape
        out = torch.empty((B, Cl, 2 * h, 2 * w), dtype=torch.float32, device=low.device)
        # max|out| for the convolution that reads the up-sampled tensor (the LDS-tiled kernel of 16-byte rows tracks it)
        am = bounds.new_amax(low.device) if bounds.enabled() and (2 * w) % 4 == 0 and out.data_ptr() % 16 == 0 else None
        with _lib.device_guard(low.device):
            rc = _lib.lib().uaps_up_cat_fwd_h(_lib.mk_hints((), am) if am is not None else None, low.data_ptr(), low.data_ptr(), out.data_ptr(), B, 0, Cl, h, w,
                                            _lib.current_stream(low.device))
        _lib.check(rc, "uaps_up_cat_fwd")
        ctx.meta = (B, Cl, h, w)
        stepctx.fwd().last_up2x_amax = am
        return out

    @staticmethod
    def backward(ctx, dout):
        B, Cl, h, w = ctx.meta
        dout = dout.contiguous()
        dlow = torch.empty((B, Cl, h, w), dtype=torch.float32, device=dout.device)
        with _lib.device_guard(dout.device):
            rc = _lib.lib().uaps_up_cat_bwd(dout.data_ptr(), None, dlow.data_ptr(), B, 0, Cl, h, w, _lib.current_stream(dout.device))
        _lib.check(rc, "uaps_up_cat_bwd")
        return dlow


def upsample2x(low: torch.Tensor) -> torch.Tensor:
    out = _Up2x.apply(low)
    f = stepctx.fwd()
    am, f.last_up2x_amax = f.last_up2x_amax, None
    return bounds.put(out, am)
