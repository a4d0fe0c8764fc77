"""Magnitude bounds of the tensors that feed the convolutions.

The fastest convolution kernels (conv mode 2, csrc/conv_split.hpp) compute an fp32 convolution from two fp16 pieces per
operand; fp16's narrow exponent needs every tensor operand scaled by a power of two into range, and the scale comes from an
upper bound of the tensor's magnitude held in DEVICE memory (the host never reads it).  A bound travels with a tensor as the
Python attribute `_uaps_bound = (bound, factor)`: |t| <= value(bound) * factor, a bound being 16 strided floats whose maximum
counts (the kernels that raise one atomically spread over the slots).  Three sources:

  * train-mode BatchNorm outputs: |gamma * x_hat + beta| <= sqrt(n) * max_c(|gamma_c| + |beta_c|) (uaps_bn_param_bounds, one
    launch per optimizer step for all layers of a model; `refresh`), passed through LeakyReLU, dropout (x 1 / (1 - p)), max-pool,
    the feature perturbations (x 1.3, x 2, x 1) and bilinear up-sampling unchanged or with their static factor;
  * kernels that produce gradients (BatchNorm backward, the loss backward) and the decoder's up-sampling raise a zeroed
    device scalar to max|output| (uaps_call_hints::out_amax);
  * anything else has no bound, and the convolution runs in the exact three-piece bf16 form instead -- correctness never
    depends on a bound being present, only on a present bound being true.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Tuple

import torch

from . import _lib

ATTR = "_uaps_bound"
Bound = Tuple[torch.Tensor, float]

_pool = {}          # (device index, stream) -> [chunk tensor, next free bound]
_CHUNK = 64
# a bound is FLOATS floats: SLOTS values STRIDE apart whose maximum counts (include/uaps_hip.h: UAPS_BOUND_*); the kernels that
# raise it atomically spread their workgroups over the slots
SLOTS, STRIDE = 16, 64
FLOATS = SLOTS * STRIDE


def get(t: Optional[torch.Tensor]) -> Optional[Bound]:
    return getattr(t, ATTR, None) if t is not None else None


def put(t: torch.Tensor, scalar: Optional[torch.Tensor], factor: float = 1.0) -> torch.Tensor:
    if scalar is not None:
        setattr(t, ATTR, (scalar, float(factor)))
    return t


def carry(src: torch.Tensor, dst: torch.Tensor, factor: float = 1.0) -> torch.Tensor:
    """dst is an elementwise contraction of src times at most `factor` (a mask, a max-pool, an interpolation ...)."""
    b = get(src)
    if b is not None:
        setattr(dst, ATTR, (b[0], b[1] * float(factor)))
    return dst


def new_amax(dev: torch.device) -> torch.Tensor:
    """A zeroed device bound for uaps_call_hints::out_amax: a view into a chunk of zeros (one fill launch per 64 bounds)."""
    key = (dev.index, _lib.current_stream(dev))      # a chunk is zero-filled on, and handed out for, one stream
    ent = _pool.get(key)
    if ent is None or ent[1] >= _CHUNK:
        # all views of a chunk are made by one call (a Python-level slice per bound costs more than the kernels' hints)
        chunk = torch.empty((_CHUNK, FLOATS), dtype=torch.float32, device=dev)
        with _lib.device_guard(dev):                # agent-scope zero fill: see uaps_zero_bounds (csrc/hints.hip)
            _lib.check(_lib.lib().uaps_zero_bounds(chunk.data_ptr(), chunk.numel(), _lib.current_stream(dev)), "uaps_zero_bounds")
        ent = _pool[key] = [chunk.unbind(0), 0]
    s = ent[0][ent[1]]
    ent[1] += 1
    return s


def from_value(v: torch.Tensor) -> torch.Tensor:
    """A bound holding the 1-element device tensor `v` (tests, and callers that computed a maximum themselves)."""
    b = torch.zeros(FLOATS, dtype=torch.float32, device=v.device)
    b[0:1] = v.reshape(1)
    return b


def value(b: torch.Tensor) -> torch.Tensor:
    return b[::STRIDE][:SLOTS].max()


def reset_pool() -> None:
    """Forget the current chunks: the next scalar comes from a fresh one (a graph capture calls this first, so that the zero
    fill of every scalar it hands to the captured kernels is part of the graph)."""
    _pool.clear()


def enabled() -> bool:
    from . import conv
    return conv.get_mode() == "h16"


# ---- BatchNorm parameter bounds, refreshed once per optimizer step -------------------------------------------------------------

def _param_stamp(bn) -> tuple:
    from . import conv
    return (conv.stamp(bn.weight), bn.weight._version, bn.bias._version)


def refresh(bns: List[torch.nn.BatchNorm2d]) -> None:
    """max_c(|gamma_c| + |beta_c|) of every layer in `bns` by one launch; each layer keeps a view of its scalar."""
    from . import conv
    if not bns or not enabled():
        return
    # current = every layer's scalar was computed for its parameters as they are: the owning optimizer's step counter (conv.stamp:
    # this package's Adam writes through raw pointers) and the tensors' version counters (load_state_dict, in-place edits)
    if all(getattr(m, "_uaps_G_gen", None) == _param_stamp(m) for m in bns) and bns[0]._uaps_G.device == bns[0].weight.device:
        return
    dev = bns[0].weight.device
    n = len(bns)
    out = torch.empty(n * FLOATS, dtype=torch.float32, device=dev)
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    with _lib.device_guard(dev):
        rc = _lib.lib().uaps_bn_param_bounds(arr([m.weight for m in bns]), arr([m.bias for m in bns]),
                                             (C.c_int * n)(*[m.num_features for m in bns]), n, out.data_ptr(), _lib.current_stream(dev))
    _lib.check(rc, "uaps_bn_param_bounds")
    for i, m in enumerate(bns):
        m._uaps_G, m._uaps_G_gen = out[i * FLOATS:(i + 1) * FLOATS], _param_stamp(m)


def bn_output_bound(bn: torch.nn.BatchNorm2d, n_per_channel: int, factor: float = 1.0) -> Optional[Bound]:
    """Bound of leaky_relu(batch_norm_train(.)) (x factor) of this layer, if its scalar is current."""
    from . import conv
    g = getattr(bn, "_uaps_G", None)
    if g is None or getattr(bn, "_uaps_G_gen", None) != _param_stamp(bn):
        return None
    return (g, math.sqrt(max(1, n_per_channel)) * float(factor))
