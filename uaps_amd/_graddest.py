"""Where the backward kernels write parameter gradients.

By default every backward allocates its gradient tensor.  The data-parallel exchange (uaps_amd.dist.GradBuckets) instead
registers, per parameter, a slice of one flat buffer per bucket: the weight-gradient / BatchNorm-backward kernels then write
straight into that slice, autograd adopts the returned view as `.grad` (a fresh view with a single reference is taken
over, not cloned), and the bucket's all-reduce runs on the flat buffer in place -- no per-step concatenation copy."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

# id(parameter) -> (flat buffer, offset in elements, shape)
_DEST: Dict[int, Tuple[torch.Tensor, int, torch.Size]] = {}


def register(param: torch.Tensor, flat: torch.Tensor, offset: int) -> None:
    _DEST[id(param)] = (flat, int(offset), param.shape)


def unregister(param: torch.Tensor) -> None:
    _DEST.pop(id(param), None)


def take(key: Optional[int], shape, device, dtype=torch.float32) -> torch.Tensor:
    """A fresh tensor for the gradient of the parameter with id `key`: a view of its registered slice when there is one
    (and it matches), else newly allocated memory."""
    ent = _DEST.get(key) if key is not None else None
    if ent is not None:
        flat, off, shp = ent
        if tuple(shp) == tuple(shape) and flat.device == device and flat.dtype == dtype:
            n = 1
            for d in shp:
                n *= int(d)
            return flat[off:off + n].view(shp)
    return torch.empty(tuple(shape), dtype=dtype, device=device)
