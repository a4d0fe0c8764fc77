"""Where the backward kernels write parameter gradients.

By default every backward allocates its gradient tensor.  The data-parallel exchange (uaps_amd.dist.GradBuckets) instead
registers, per parameter, a slice of one flat buffer per bucket: the weight-gradient / BatchNorm-backward kernels then write
straight into that slice, autograd adopts the returned view as `.grad` (a fresh view with a single reference is taken
over, not cloned), and the bucket's all-reduce runs on the flat buffer in place -- no per-step concatenation copy."""
from __future__ import annotations

from typing import Dict, Optional, Set, Tuple

import torch

# id(parameter) -> (flat buffer, offset in elements, shape)
_DEST: Dict[int, Tuple[torch.Tensor, int, torch.Size]] = {}
# ids whose slice was handed out since the last new_backward(): a parameter used by SEVERAL autograd nodes of one backward
# (two forwards of one model: a ragged last batch, pair_forward=False, a custom loss) gets its slice once and freshly allocated
# memory afterwards -- autograd then sums the two tensors, and GradBuckets._launch copies a sum that landed elsewhere into
# the slice.  Handing the slice out twice would let the second node overwrite the first node's gradient (2 g2, not g1 + g2).
_TAKEN: Set[int] = set()


def register(param: torch.Tensor, flat: torch.Tensor, offset: int) -> None:
    _DEST[id(param)] = (flat, int(offset), param.shape)


def unregister(param: torch.Tensor) -> None:
    _DEST.pop(id(param), None)
    _TAKEN.discard(id(param))


def new_backward() -> None:
    """Every registered slice may be handed out again (GradBuckets.reset: once per step, after the exchange)."""
    _TAKEN.clear()


def take(key: Optional[int], shape, device, dtype=torch.float32) -> torch.Tensor:
    """A fresh tensor for the gradient of the parameter with id `key`: a view of its registered slice when there is one
    (it matches, and this is the parameter's first gradient of this backward), else newly allocated memory."""
    ent = _DEST.get(key) if key is not None else None
    if ent is not None and key not in _TAKEN:
        flat, off, shp = ent
        if tuple(shp) == tuple(shape) and flat.device == device and flat.dtype == dtype:
            n = 1
            for d in shp:
                n *= int(d)
            _TAKEN.add(key)
            return flat[off:off + n].view(shp)
    return torch.empty(tuple(shape), dtype=dtype, device=device)
