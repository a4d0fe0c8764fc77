"""Host side of the convolution kernels (csrc/conv_kernels.hpp, C ABI section "Convolutions").

`conv2d(x, weight, bias)` is nn.Conv2d's forward for the only two forms the reference U-Net uses
(3x3 / padding 1 and 1x1, stride 1: utilities/UAPS_unet.py:37,41,73,138) with a hand-written
backward: the input gradient is the same implicit-GEMM kernel on the transposed, tap-flipped packed
weights, the weight (and bias) gradient a pixel-split MFMA reduction.  There is no MIOpen / PyTorch
fallback on this path: CPU tensors are refused.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, Optional, Tuple

import torch
from torch.optim.optimizer import register_optimizer_step_post_hook

from . import _lib

_ws: Dict[Tuple[int, int], torch.Tensor] = {}
# packed weights per parameter: id(weight) -> (weakref, version, generation, wf, wb)
_packed: Dict[int, tuple] = {}
# Bumped after every torch.optim optimizer step (global post-step hook): fused / foreach optimizers
# update parameters without touching Tensor._version, so the version counter alone is not enough.
_generation = 0


def invalidate_packed_weights(*_args, **_kwargs) -> None:
    """Forget every packed weight buffer.  Runs automatically after each `optimizer.step()` of any
    torch.optim optimizer; call it yourself after writing to parameters through `.data` or other
    paths that bypass both the optimizer and the tensor version counter."""
    global _generation
    _generation += 1


register_optimizer_step_post_hook(invalidate_packed_weights)


# bench.py sets KERNEL_EVENTS to a dict {kernel instantiation name: [(start_event, end_event, flops), ...]}
# to time individual launches with HIP events on the stream they run on; EVENT_FILTER (a set of names)
# restricts the recording to those instantiations.  None (the default) records nothing.
KERNEL_EVENTS: Optional[Dict[str, list]] = None
EVENT_FILTER: Optional[set] = None
_variant_cache: Dict[tuple, str] = {}


def kernel_variant(kind: str, B: int, Cin: int, Cout: int, H: int, W: int, ks: int, cfg: int = 0) -> str:
    """Name of the kernel instantiation a call launches ('fwd' / 'bwd_data' / 'wrw'), as rocprofv3 prints it."""
    key = (kind, B, Cin, Cout, H, W, ks, cfg)
    name = _variant_cache.get(key)
    if name is None:
        buf = C.create_string_buffer(96)
        L = _lib.lib()
        if kind == "wrw":
            rc = L.uaps_conv_wrw_variant(B, Cin, Cout, H, W, ks, cfg, buf, 96)
        elif kind == "bwd_data":
            rc = L.uaps_conv_fwd_variant(B, Cout, Cin, H, W, ks, cfg, buf, 96)
        else:
            rc = L.uaps_conv_fwd_variant(B, Cin, Cout, H, W, ks, cfg, buf, 96)
        _lib.check(rc, "uaps_conv_*_variant")
        name = _variant_cache[key] = buf.value.decode()
    return name


class _timed:
    def __init__(self, kind, B, Cin, Cout, H, W, ks, cfg):
        self.on = False
        if KERNEL_EVENTS is not None:
            self.name = kernel_variant(kind, B, Cin, Cout, H, W, ks, cfg)
            self.on = EVENT_FILTER is None or self.name in EVENT_FILTER
            self.flops = 2.0 * B * H * W * Cin * Cout * ks * ks

    def __enter__(self):
        if self.on:
            self.s, self.e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if self.on:
            self.e.record()
            KERNEL_EVENTS.setdefault(self.name, []).append((self.s, self.e, self.flops))
        return False


def _workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    key = (dev.index, _lib.current_stream(dev))
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8, device=dev)
        _ws[key] = w
    return w


def pack_weights(weight: torch.Tensor, need_bwd: bool = True):
    """Packed forward / input-gradient weight buffers of `weight` [Cout,Cin,ks,ks], cached until the
    parameter is modified in place (optimizer step) or replaced."""
    key = id(weight)
    ent = _packed.get(key)
    if ent is not None:
        ref, ver, gen, wf, wb = ent
        if (ref() is weight and ver == weight._version and gen == _generation and wf.device == weight.device
                and (wb is not None or not need_bwd)):
            return wf, wb
    Cout, Cin, ks, ks2 = weight.shape
    if ks != ks2 or ks not in (1, 3):
        raise ValueError(f"conv2d: kernel {ks}x{ks2} not supported (the U-Net uses 3x3 and 1x1)")
    nf, nb = C.c_size_t(), C.c_size_t()
    L = _lib.lib()
    _lib.check(L.uaps_conv_pack_floats(Cout, Cin, ks, C.byref(nf), C.byref(nb)), "uaps_conv_pack_floats")
    dev = weight.device
    wf = torch.empty(nf.value, dtype=torch.float32, device=dev)
    wb = torch.empty(nb.value, dtype=torch.float32, device=dev) if need_bwd else None
    w = weight.detach().contiguous()
    with torch.cuda.device(dev):
        rc = L.uaps_conv_pack_weights(w.data_ptr(), Cout, Cin, ks, wf.data_ptr(), wb.data_ptr() if need_bwd else None,
                                      _lib.current_stream(dev))
    _lib.check(rc, "uaps_conv_pack_weights")
    if len(_packed) > 4096:
        for k in [k for k, e in _packed.items() if e[0]() is None]:
            del _packed[k]
    _packed[key] = (weakref.ref(weight), weight._version, _generation, wf, wb)
    return wf, wb


def conv_fwd_raw(x: torch.Tensor, wf: torch.Tensor, bias: Optional[torch.Tensor], Cout: int, ks: int, cfg: int = 0):
    B, Cin, H, W = x.shape
    y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timed("fwd", B, Cin, Cout, H, W, ks, cfg):
        rc = _lib.lib().uaps_conv_fwd(x.data_ptr(), wf.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                      B, Cin, Cout, H, W, ks, cfg, _lib.current_stream(x.device))
    _lib.check(rc, "uaps_conv_fwd")
    return y


def conv_bwd_data_raw(dy: torch.Tensor, wb: torch.Tensor, Cin: int, ks: int, cfg: int = 0):
    B, Cout, H, W = dy.shape
    dx = torch.empty((B, Cin, H, W), dtype=torch.float32, device=dy.device)
    with torch.cuda.device(dy.device), _timed("bwd_data", B, Cin, Cout, H, W, ks, cfg):
        rc = _lib.lib().uaps_conv_bwd_data(dy.data_ptr(), wb.data_ptr(), dx.data_ptr(), B, Cin, Cout, H, W, ks, cfg,
                                           _lib.current_stream(dy.device))
    _lib.check(rc, "uaps_conv_bwd_data")
    return dx


def conv_bwd_weight_raw(dy: torch.Tensor, x: torch.Tensor, ks: int, want_bias: bool, cfg: int = 0):
    B, Cout, H, W = dy.shape
    Cin = x.shape[1]
    dev = dy.device
    L = _lib.lib()
    n = C.c_size_t()
    _lib.check(L.uaps_conv_wrw_workspace_bytes(B, Cin, Cout, H, W, ks, cfg, C.byref(n)), "uaps_conv_wrw_workspace_bytes")
    ws = _workspace(dev, n.value)
    dw = torch.empty((Cout, Cin, ks, ks), dtype=torch.float32, device=dev)
    db = torch.empty(Cout, dtype=torch.float32, device=dev) if want_bias else None
    with torch.cuda.device(dev):
        st = _lib.current_stream(dev)
        with _timed("wrw", B, Cin, Cout, H, W, ks, cfg):
            rc = L.uaps_conv_bwd_weight_partial(dy.data_ptr(), x.data_ptr(), int(want_bias), B, Cin, Cout, H, W, ks, cfg,
                                                ws.data_ptr(), ws.numel(), st)
        _lib.check(rc, "uaps_conv_bwd_weight_partial")
        rc = L.uaps_conv_bwd_weight_reduce(ws.data_ptr(), dw.data_ptr(), db.data_ptr() if want_bias else None, B, Cin, Cout, H, W,
                                           ks, cfg, st)
    _lib.check(rc, "uaps_conv_bwd_weight_reduce")
    return dw, db


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        _lib.require_device(x, "conv2d")
        if x.dtype != torch.float32 or weight.dtype != torch.float32:
            raise TypeError("conv2d: fp32 only (the reference trains in fp32)")
        x = x.contiguous()
        Cout, Cin, ks, _ = weight.shape
        if x.shape[1] != Cin:
            raise ValueError(f"conv2d: input has {x.shape[1]} channels, weight expects {Cin}")
        need_bwd = ctx.needs_input_grad[0]
        wf, wb = pack_weights(weight, need_bwd=True)
        y = conv_fwd_raw(x, wf, bias, Cout, ks)
        ctx.save_for_backward(x, wb)
        ctx.meta = (Cin, Cout, ks, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wb = ctx.saved_tensors
        Cin, Cout, ks, has_bias = ctx.meta
        dy = dy.contiguous()
        dx = conv_bwd_data_raw(dy, wb, Cin, ks) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dw, db = conv_bwd_weight_raw(dy, x, ks, has_bias and ctx.needs_input_grad[2])
        return dx, dw, db


def conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """F.conv2d(x, weight, bias, stride=1, padding=weight.shape[-1] // 2) for 3x3 and 1x1 kernels."""
    return _Conv2d.apply(x, weight, bias)
