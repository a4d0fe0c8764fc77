"""Feature perturbations of the three auxiliary decoders (reference utilities/UAPS_unet.py:156-185),
as HIP kernels with a counter-based on-device RNG (csrc/perturb.hip).

Names and call forms follow the reference: `FeatureNoise()(x)`, `Dropout(x)`, `FeatureDropout(x)`.
The reference draws FeatureNoise / Dropout from the torch CPU/GPU generators and the FeatureDropout
threshold from numpy's global RNG; here noise and keep-masks come from Philox4x32-10 keyed by
`manual_seed()` (+ rank), regenerated in the backward instead of stored, and the FeatureDropout
threshold still comes from numpy's global RNG (a host scalar, as in the reference).  The `*_with`
forms take recorded draws, for parity tests against the reference.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib, bounds, config, stepctx
from . import conv as _conv


# Captured steps (trainer graph mode) cannot take a fresh host number per replay: with DEVICE_THRESHOLDS the FeatureDropout
# threshold factors U(0.7, 0.9) (UAPS_unet.py:164) are drawn by the fan-out kernel itself from its own Philox counter.
DEVICE_THRESHOLDS = False


class RngState:
    """(seed, running Philox counter offset) of the perturbation draws."""

    def __init__(self, seed: int = 0x5EED_0A95, offset: int = 0):
        self.seed, self.offset = int(seed), int(offset)

    def reserve(self, n_elements: int) -> Tuple[int, int]:
        off = self.offset
        self.offset += (n_elements + 3) // 4 + 1
        return self.seed, off


# One stream per process (ranks use different seeds), as the reference has one torch / numpy global RNG.  A thread that drives a
# model of its own beside other threads can give itself a private stream (`local_rng`): the draws of its forwards then do not
# depend on how the threads interleave (tests/test_gpu_threads.py).
_RngState = RngState()


def rng() -> RngState:
    """The stream the calling thread's forward draws from: its private one inside `local_rng`, else the process's."""
    return stepctx.fwd().rng or _RngState


def _mix(seed: int, rank: int) -> int:
    return (int(seed) * 0x9E3779B97F4A7C15 + int(rank) * 0xD1B54A32D192ED03 + 1) & 0xFFFFFFFFFFFFFFFF


def manual_seed(seed: int, rank: int = 0) -> None:
    r = rng()
    r.seed, r.offset = _mix(seed, rank), 0


class local_rng:
    """`with local_rng(seed, rank):` -- the calling thread's forwards draw their perturbations from a private stream (the one
    manual_seed(seed, rank) would start) until the block ends."""

    def __init__(self, seed: int, rank: int = 0):
        self.state = RngState(_mix(seed, rank), 0)

    def __enter__(self):
        f = stepctx.fwd()
        self.prev, f.rng = f.rng, self.state
        return self.state

    def __exit__(self, *a):
        stepctx.fwd().rng = self.prev
        return False


def get_rng_state() -> Tuple[int, int]:
    r = rng()
    return r.seed, r.offset


def set_rng_state(state: Tuple[int, int]) -> None:
    r = rng()
    r.seed, r.offset = int(state[0]), int(state[1])


def _prep(x: torch.Tensor, what: str) -> torch.Tensor:
    _lib.require_device(x, what)
    if x.dtype != torch.float32:
        raise TypeError(f"{what}: float32 features expected, got {x.dtype}")
    if x.dim() != 4:
        raise ValueError(f"{what}: [B,C,H,W] expected, got {tuple(x.shape)}")
    return x.contiguous()


# ---- FeatureNoise -------------------------------------------------------------------------------

class _NoiseRng(torch.autograd.Function):
    """`offsets` holds one Philox offset per group of B/len(offsets) consecutive images: every group gets its own
    [C,H,W] noise tensor, as one FeatureNoise call per reference forward would draw."""

    @staticmethod
    def forward(ctx, x, seed, offsets, rng, want_noise):
        ctx.set_materialize_grads(False)
        x = _prep(x, "FeatureNoise")
        B, Cc, H, W = x.shape
        G = len(offsets)
        if B % G:
            raise ValueError(f"FeatureNoise: batch {B} is not divisible into {G} groups")
        Bg, chw = B // G, Cc * H * W
        y = torch.empty_like(x)
        noise = torch.empty((G, Cc, H, W), dtype=torch.float32, device=x.device) if want_noise else None
        with _lib.device_guard(x.device):
            for g, off in enumerate(offsets):
                rc = _lib.lib().uaps_feat_noise(x.data_ptr() + 4 * g * Bg * chw, y.data_ptr() + 4 * g * Bg * chw, Bg, Cc, H, W,
                                                seed, off, float(rng), noise[g].data_ptr() if want_noise else None,
                                                _lib.current_stream(x.device))
                _lib.check(rc, "uaps_feat_noise")
        ctx.meta = (seed, tuple(offsets), float(rng))
        if want_noise:
            noise = noise[0] if G == 1 else noise
            ctx.mark_non_differentiable(noise)
            return y, noise
        return y

    @staticmethod
    def backward(ctx, gy, *unused):
        seed, offsets, rng = ctx.meta
        if gy is None:
            return None, None, None, None, None
        gy = gy.contiguous()
        B, Cc, H, W = gy.shape
        Bg, chw = B // len(offsets), Cc * H * W
        gx = torch.empty_like(gy)
        with _lib.device_guard(gy.device):
            for g, off in enumerate(offsets):
                rc = _lib.lib().uaps_feat_noise(gy.data_ptr() + 4 * g * Bg * chw, gx.data_ptr() + 4 * g * Bg * chw, Bg, Cc, H, W,
                                                seed, off, rng, None, _lib.current_stream(gy.device))
                _lib.check(rc, "uaps_feat_noise (backward)")
        return gx, None, None, None, None


class _NoiseApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, noise):
        x = _prep(x, "feature_noise_with")
        noise = noise.to(torch.float32).contiguous()
        if tuple(noise.shape) != tuple(x.shape[1:]):
            raise ValueError("noise must have the per-sample shape [C,H,W]")
        y = torch.empty_like(x)
        with _lib.device_guard(x.device):
            rc = _lib.lib().uaps_feat_noise_apply(x.data_ptr(), noise.data_ptr(), y.data_ptr(), x.shape[0],
                                                  noise.numel(), _lib.current_stream(x.device))
        _lib.check(rc, "uaps_feat_noise_apply")
        ctx.save_for_backward(noise)
        return y

    @staticmethod
    def backward(ctx, gy):
        (noise,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(gy)
        with _lib.device_guard(gy.device):
            rc = _lib.lib().uaps_feat_noise_apply(gy.data_ptr(), noise.data_ptr(), gx.data_ptr(), gy.shape[0],
                                                  noise.numel(), _lib.current_stream(gy.device))
        _lib.check(rc, "uaps_feat_noise_apply (backward)")
        return gx, None


def feature_noise_with(x: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """x*noise + x with a given [C,H,W] noise tensor (UAPS_unet.py:177-181 with the draw recorded)."""
    return _NoiseApply.apply(x, noise)


class FeatureNoise(nn.Module):
    """UAPS_unet.py:172-185: x * n + x, n ~ U(-r, r) of shape [C,H,W], shared by the batch."""

    def __init__(self, uniform_range: float = 0.3):
        super().__init__()
        self.uniform_range = float(uniform_range)

    def forward(self, x: torch.Tensor, return_noise: bool = False, groups: int = 1):
        offs = []
        for _ in range(groups):
            seed, off = rng().reserve(x[0].numel())
            offs.append(off)
        return _NoiseRng.apply(x, seed, tuple(offs), self.uniform_range, return_noise)


# ---- Dropout(x, p=0.5), always in training mode ---------------------------------------------------

class _Bernoulli(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, seed, offset, p, want_keep):
        ctx.set_materialize_grads(False)
        x = _prep(x, "Dropout")
        y = torch.empty_like(x)
        keep = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if want_keep else None
        with _lib.device_guard(x.device):
            rc = _lib.lib().uaps_feat_bernoulli(x.data_ptr(), y.data_ptr(), x.numel(), seed, offset, float(p),
                                                keep.data_ptr() if want_keep else None, _lib.current_stream(x.device))
        _lib.check(rc, "uaps_feat_bernoulli")
        ctx.meta = (seed, offset, float(p))
        if want_keep:
            ctx.mark_non_differentiable(keep)
            return y, keep
        return y

    @staticmethod
    def backward(ctx, gy, *unused):
        seed, offset, p = ctx.meta
        if gy is None:
            return None, None, None, None, None
        gy = gy.contiguous()
        gx = torch.empty_like(gy)
        with _lib.device_guard(gy.device):
            rc = _lib.lib().uaps_feat_bernoulli(gy.data_ptr(), gx.data_ptr(), gy.numel(), seed, offset, p, None,
                                                _lib.current_stream(gy.device))
        _lib.check(rc, "uaps_feat_bernoulli (backward)")
        return gx, None, None, None, None


class _MaskApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, keep, scale):
        x = _prep(x, "feature_mask_with")
        keep = keep.to(torch.uint8).contiguous()
        if keep.shape != x.shape:
            raise ValueError("keep mask must have the shape of x")
        y = torch.empty_like(x)
        with _lib.device_guard(x.device):
            rc = _lib.lib().uaps_feat_mask_apply(x.data_ptr(), keep.data_ptr(), float(scale), y.data_ptr(), x.numel(),
                                                 _lib.current_stream(x.device))
        _lib.check(rc, "uaps_feat_mask_apply")
        ctx.save_for_backward(keep)
        ctx.scale = float(scale)
        return y

    @staticmethod
    def backward(ctx, gy):
        (keep,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(gy)
        with _lib.device_guard(gy.device):
            rc = _lib.lib().uaps_feat_mask_apply(gy.data_ptr(), keep.data_ptr(), ctx.scale, gx.data_ptr(), gy.numel(),
                                                 _lib.current_stream(gy.device))
        _lib.check(rc, "uaps_feat_mask_apply (backward)")
        return gx, None, None


def dropout_with(x: torch.Tensor, keep: torch.Tensor, p: float = 0.5) -> torch.Tensor:
    """F.dropout(x, p, training=True) with the keep mask recorded: x * keep / (1-p)."""
    return _MaskApply.apply(x, keep, 1.0 / (1.0 - p))


def Dropout(x: torch.Tensor, p: float = 0.5, return_keep: bool = False):
    """UAPS_unet.py:156-158: F.dropout(x, p) with training=True unconditionally (also in eval())."""
    seed, off = rng().reserve(x.numel())
    return _Bernoulli.apply(x, seed, off, p, return_keep)


# ---- FeatureDropout -------------------------------------------------------------------------------

_fd_ws: Dict[Tuple[int, int], torch.Tensor] = {}


class _FeatDrop(torch.autograd.Function):
    """`u` is one threshold factor, or a tuple with one per group of B/len(u) consecutive images."""

    @staticmethod
    def forward(ctx, x, u):
        ctx.set_materialize_grads(False)
        x = _prep(x, "FeatureDropout")
        B, Cc, H, W = x.shape
        us = tuple(u) if isinstance(u, (tuple, list)) else (float(u),)
        G = len(us)
        if B % G:
            raise ValueError(f"FeatureDropout: batch {B} is not divisible into {G} groups")
        Bg = B // G
        L = _lib.lib()
        need = C.c_size_t()
        _lib.check(L.uaps_feat_dropout_workspace_bytes(Bg, Cc, H, W, C.byref(need)), "uaps_feat_dropout_workspace_bytes")
        key = (x.device.index, _lib.current_stream(x.device))
        ws = _fd_ws.get(key)
        if ws is None or ws.numel() < need.value:
            ws = torch.empty(need.value, dtype=torch.uint8, device=x.device)
            _fd_ws[key] = ws
        y = torch.empty_like(x)
        keep = torch.empty((B, H, W), dtype=torch.uint8, device=x.device)
        with _lib.device_guard(x.device):
            for g, ug in enumerate(us):
                o = g * Bg * Cc * H * W * 4
                rc = L.uaps_feat_dropout_fwd(x.data_ptr() + o, y.data_ptr() + o, Bg, Cc, H, W, float(ug),
                                             keep.data_ptr() + g * Bg * H * W, ws.data_ptr(), ws.numel(),
                                             _lib.current_stream(x.device))
                _lib.check(rc, "uaps_feat_dropout_fwd")
        ctx.save_for_backward(keep)
        ctx.mark_non_differentiable(keep)
        return y, keep

    @staticmethod
    def backward(ctx, gy, _gk):
        (keep,) = ctx.saved_tensors
        if gy is None:
            return None, None
        gy = gy.contiguous()
        B, Cc, H, W = gy.shape
        gx = torch.empty_like(gy)
        with _lib.device_guard(gy.device):
            rc = _lib.lib().uaps_feat_dropout_bwd(gy.data_ptr(), keep.data_ptr(), gx.data_ptr(), B, Cc, H, W,
                                                  _lib.current_stream(gy.device))
        _lib.check(rc, "uaps_feat_dropout_bwd")
        return gx, None


def feature_dropout_with(x: torch.Tensor, u: float, return_keep: bool = False):
    """UAPS_unet.py:161-169 with the np.random.uniform(0.7, 0.9) draw passed in."""
    y, keep = _FeatDrop.apply(x, tuple(u) if isinstance(u, (tuple, list)) else float(u))
    return (y, keep) if return_keep else y


def FeatureDropout(x: torch.Tensor, groups: int = 1) -> torch.Tensor:
    """UAPS_unet.py:161-169: zero the pixels whose channel-mean reaches U(0.7,0.9) x the sample maximum
    (one threshold draw per call of the reference, i.e. per group here)."""
    if groups == 1:
        return feature_dropout_with(x, np.random.uniform(0.7, 0.9))
    return feature_dropout_with(x, tuple(np.random.uniform(0.7, 0.9) for _ in range(groups)))


# ---- all decoders' views of one encoder feature map, with a fused backward ---------------------------------

_KIND_MODE = {"noise": 1, "dropout": 2, "feature_dropout": 3}
# (stepctx.fwd().fan_side, set by UNet_UAPS.forward around its encoder loop: the stream the perturbed copies are written on; None: the caller's)
_FUSED_FANOUT = config.flag("UAPS_FUSED_FANOUT", True)      # A/B switch for tools/ab_bench.sh


class _PerturbFan(torch.autograd.Function):
    """(f, kinds) -> (f, P_1(f), ..., P_n(f)): the clean feature map for the main decoder and one perturbed copy per
    auxiliary decoder (UAPS_unet.py:226-232).  The forward launches the same kernels as FeatureNoise / Dropout /
    FeatureDropout; the backward is ONE kernel (uaps_fanin_perturbed) that re-applies each perturbation to its
    incoming gradient and sums, instead of a backward kernel per perturbation plus a fan-in sum."""

    @staticmethod
    def forward(ctx, f, kinds, groups, noise_range, drop_p, with_pool=False):
        ctx.set_materialize_grads(False)
        ctx.step = stepctx.current()       # the trainer scope this forward runs in: the backward starts its early flush (conv.early_flush)
        f = _prep(f, "perturbed feature fan-out")
        B, Cc, H, W = f.shape
        if B % groups:
            raise ValueError(f"batch {B} is not divisible into {groups} groups")
        Bg, chw, dev = B // groups, Cc * H * W, f.device
        L = _lib.lib()
        outs, offsets, keeps = [f.view_as(f)], [[0] * groups], [None]
        seed = rng().seed
        n = len(kinds)
        # the one-pass kernel carries one set of FeatureDropout thresholds: with two FeatureDropout decoders (n_aux >= 6) the
        # per-perturbation kernels below run instead (decided before any random number is reserved)
        if _FUSED_FANOUT and n >= 1 and (H * W) % 4 == 0 and groups <= 4 and n <= 8 and f.data_ptr() % 16 == 0 \
                and all(k in _KIND_MODE for k in kinds) and sum(k == "feature_dropout" for k in kinds) <= 1:
            # one pass over f for all the copies (same draws, same order of RNG reservations as the per-kernel path below)
            ys = [torch.empty_like(f) for _ in kinds]
            us = [0.0] * groups
            kp = [None] * n
            ws = None
            for i, kind in enumerate(kinds):
                if kind == "noise":
                    offsets.append([rng().reserve(chw)[1] for _ in range(groups)]); keeps.append(None)
                elif kind == "dropout":
                    offsets.append([rng().reserve(f.numel())[1]] * groups); keeps.append(None)
                else:
                    need = C.c_size_t()
                    _lib.check(L.uaps_feat_dropout_workspace_bytes(B, Cc, H, W, C.byref(need)), "uaps_feat_dropout_workspace_bytes")
                    key = (dev.index, _lib.current_stream(dev))
                    ws = _fd_ws.get(key)
                    if ws is None or ws.numel() < need.value:
                        ws = _fd_ws[key] = torch.empty(need.value, dtype=torch.uint8, device=dev)
                    kp[i] = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
                    if DEVICE_THRESHOLDS:
                        us = [-1.0] * groups
                        offsets.append([rng().reserve(4)[1] for _ in range(groups)]); keeps.append(kp[i])
                    else:
                        us = [float(np.random.uniform(0.7, 0.9)) for _ in range(groups)]
                        offsets.append([0] * groups); keeps.append(kp[i])
            with _lib.device_guard(dev):
                st = _lib.current_stream(dev)
                pooled = idx = None
                if with_pool:                # first: the next encoder level waits for this alone
                    pooled = torch.empty((B, Cc, H // 2, W // 2), dtype=torch.float32, device=dev)
                    idx = torch.empty((B, Cc, H // 2, W // 2), dtype=torch.uint8, device=dev)
                    _lib.check(L.uaps_maxpool2x2_fwd(f.data_ptr(), B, Cc, H, W, pooled.data_ptr(), idx.data_ptr(), st), "uaps_maxpool2x2_fwd")
                # the perturbed copies are read by the decoders only: with a side stream set (UNet_UAPS.forward, decoder-stream mode) they
                # are written beside the encoder's next levels, and the caller makes the decoders wait for that stream
                fs = stepctx.fwd().fan_side
                side = fs if (fs is not None and fs.device == dev) else None
                if side is not None:
                    side.wait_stream(torch.cuda.current_stream(dev))
                with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                    st2 = _lib.current_stream(dev)
                    if ws is not None:
                        _lib.check(L.uaps_feat_dropout_stats(f.data_ptr(), B, Cc, H, W, ws.data_ptr(), ws.numel(), st2), "uaps_feat_dropout_stats")
                    rc = L.uaps_fanout_perturbed(f.data_ptr(), (C.c_void_p * n)(*[y.data_ptr() for y in ys]),
                                                 (C.c_int * n)(*[_KIND_MODE[k] for k in kinds]),
                                                 (C.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in kp]),
                                                 (C.c_uint64 * (n * groups))(*[o for offs in offsets[1:] for o in offs]),
                                                 (C.c_float * groups)(*us), ws.data_ptr() if ws is not None else None, n, groups, seed,
                                                 float(noise_range), float(drop_p), B, Cc, H, W, st2)
                    _lib.check(rc, "uaps_fanout_perturbed")
                outs.extend(ys)
                if with_pool:
                    outs.append(pooled); offsets.append([0] * groups); keeps.append(idx)
            ctx.meta = (tuple(kinds), groups, seed, float(noise_range), float(drop_p), offsets, (B, Cc, H, W), bool(with_pool))
            ctx.keeps = keeps
            return tuple(outs)
        with _lib.device_guard(dev):
            st = _lib.current_stream(dev)
            for kind in kinds:
                y = torch.empty_like(f)
                if kind == "noise":
                    offs = [rng().reserve(chw)[1] for _ in range(groups)]
                    for g, off in enumerate(offs):
                        _lib.check(L.uaps_feat_noise(f.data_ptr() + 4 * g * Bg * chw, y.data_ptr() + 4 * g * Bg * chw, Bg, Cc, H, W, seed,
                                                     off, float(noise_range), None, st), "uaps_feat_noise")
                    offsets.append(offs); keeps.append(None)
                elif kind == "dropout":
                    off = rng().reserve(f.numel())[1]
                    _lib.check(L.uaps_feat_bernoulli(f.data_ptr(), y.data_ptr(), f.numel(), seed, off, float(drop_p), None, st),
                               "uaps_feat_bernoulli")
                    offsets.append([off] * groups); keeps.append(None)
                elif kind == "feature_dropout":
                    need = C.c_size_t()
                    _lib.check(L.uaps_feat_dropout_workspace_bytes(Bg, Cc, H, W, C.byref(need)), "uaps_feat_dropout_workspace_bytes")
                    key = (dev.index, st)
                    ws = _fd_ws.get(key)
                    if ws is None or ws.numel() < need.value:
                        ws = _fd_ws[key] = torch.empty(need.value, dtype=torch.uint8, device=dev)
                    keep = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
                    for g in range(groups):
                        o = g * Bg * chw * 4
                        _lib.check(L.uaps_feat_dropout_fwd(f.data_ptr() + o, y.data_ptr() + o, Bg, Cc, H, W, float(np.random.uniform(0.7, 0.9)),
                                                           keep.data_ptr() + g * Bg * H * W, ws.data_ptr(), ws.numel(), st),
                                   "uaps_feat_dropout_fwd")
                    offsets.append([0] * groups); keeps.append(keep)
                else:
                    raise ValueError(f"unknown perturbation {kind!r}")
                outs.append(y)
            if with_pool:                    # MaxPool2d(2) of the next DownBlock (UAPS_unet.py:55-58), arg-max kept for the backward
                pooled = torch.empty((B, Cc, H // 2, W // 2), dtype=torch.float32, device=dev)
                idx = torch.empty((B, Cc, H // 2, W // 2), dtype=torch.uint8, device=dev)
                _lib.check(L.uaps_maxpool2x2_fwd(f.data_ptr(), B, Cc, H, W, pooled.data_ptr(), idx.data_ptr(), st), "uaps_maxpool2x2_fwd")
                outs.append(pooled); offsets.append([0] * groups); keeps.append(idx)
        ctx.meta = (tuple(kinds), groups, seed, float(noise_range), float(drop_p), offsets, (B, Cc, H, W), bool(with_pool))
        ctx.keeps = keeps
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        kinds, groups, seed, rng, p, offsets, (B, Cc, H, W), with_pool = ctx.meta
        modes = [0] + [_KIND_MODE[k] for k in kinds] + ([4] if with_pool else [])
        live = [(g.contiguous(), modes[i], offsets[i], ctx.keeps[i]) for i, g in enumerate(grads) if g is not None]
        if not live:
            return None, None, None, None, None, None
        _conv.early_flush(ctx.step)         # (a training step's scope: the decoders' weight-gradient reductions start beside the encoder's backward)
        if live[0][1] == 4:                 # the output buffer takes its shape from the first entry: keep a full-size one first
            live.append(live.pop(0))
        dev = live[0][0].device
        out = torch.empty_like(live[0][0])
        n = len(live)
        L = _lib.lib()
        fused_ok = (H * W) % 4 == 0 and groups <= 4 and n <= 8 and all(t[0].data_ptr() % 16 == 0 for t in live)
        with _lib.device_guard(dev):
            st = _lib.current_stream(dev)
            if fused_ok:
                gp = (C.c_void_p * n)(*[t[0].data_ptr() for t in live])
                md = (C.c_int * n)(*[t[1] for t in live])
                kp = (C.c_void_p * n)(*[(t[3].data_ptr() if t[3] is not None else None) for t in live])
                of = (C.c_uint64 * (n * groups))(*[o for t in live for o in t[2]])
                _lib.check(L.uaps_fanin_perturbed(gp, md, kp, of, n, groups, seed, rng, p, B, Cc, H, W, out.data_ptr(), st),
                           "uaps_fanin_perturbed")
                return out, None, None, None, None, None
            if any(t[1] == 4 for t in live):
                raise _lib.UapsHipError("fused max-pool backward needs H*W % 4 == 0 and 16-byte aligned gradients")
            # general shapes: one backward kernel per perturbation, then the plain fan-in sum
            Bg, chw = B // groups, Cc * H * W
            parts = []
            for g, mode, offs, keep in live:
                if mode == 0:
                    parts.append(g)
                    continue
                d = torch.empty_like(g)
                if mode == 1:
                    for q, off in enumerate(offs):
                        _lib.check(L.uaps_feat_noise(g.data_ptr() + 4 * q * Bg * chw, d.data_ptr() + 4 * q * Bg * chw, Bg, Cc, H, W, seed, off,
                                                     rng, None, st), "uaps_feat_noise (backward)")
                elif mode == 2:
                    _lib.check(L.uaps_feat_bernoulli(g.data_ptr(), d.data_ptr(), g.numel(), seed, offs[0], p, None, st),
                               "uaps_feat_bernoulli (backward)")
                else:
                    _lib.check(L.uaps_feat_dropout_bwd(g.data_ptr(), keep.data_ptr(), d.data_ptr(), B, Cc, H, W, st), "uaps_feat_dropout_bwd")
                parts.append(d)
            acc = parts
            while len(acc) > 1:
                chunk, acc = acc[:4], acc[4:]
                ptrs = (C.c_void_p * len(chunk))(*[t.data_ptr() for t in chunk])
                _lib.check(L.uaps_sum_tensors(ptrs, len(chunk), out.data_ptr(), out.numel(), st), "uaps_sum_tensors")
                acc = [out] + acc
            return (out if len(parts) > 1 else parts[0]), None, None, None, None, None


def perturbed_fan_out(f: torch.Tensor, kinds, groups: int = 1, noise_range: float = 0.3, drop_p: float = 0.5,
                      with_pool: bool = False):
    """[f, P_1(f), ..., P_n(f)] for kinds in {"noise", "dropout", "feature_dropout"} (one entry per auxiliary decoder);
    with_pool appends MaxPool2d(2)(f), whose gradient returns through the same fused backward kernel."""
    if with_pool and (f.shape[2] % 2 or f.shape[3] % 8):
        raise ValueError("with_pool needs an even height and a width that is a multiple of 8")
    outs = _PerturbFan.apply(f, tuple(kinds), int(groups), noise_range, drop_p, bool(with_pool))
    if bounds.get(f) is not None:            # |f (1 + U(-r, r))| <= (1 + r) |f|, dropout scales by 1 / (1 - p), masks and max-pool contract
        factor = {"noise": 1.0 + abs(float(noise_range)), "dropout": 1.0 / (1.0 - float(drop_p)), "feature_dropout": 1.0}
        fs = [1.0] + [factor[k] for k in kinds] + ([1.0] if with_pool else [])
        for o, m in zip(outs, fs):
            bounds.carry(f, o, m)
    return outs
