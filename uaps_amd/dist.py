"""Data-parallel gradient exchange for the UAPS step: one process per GPU, RCCL over xGMI.

The reference only has single-process nn.DataParallel (UAPS_model.py:13), whose backward does an
implicit reduce-add of the 208 gradient tensors onto GPU 0.  Here every rank runs the whole step
on its own shard of the labelled + unlabelled batch and the 3.7 M gradients (14.9 MB fp32) are
averaged with a handful of large all-reduces: xGMI is point-to-point and a 15 MB ring all-reduce
is latency-bound, so fewer, larger messages win.  Buckets follow the order in
which backward finishes them (auxiliary decoders, main decoder, encoder last) and each bucket's
all-reduce is issued from a post-accumulate hook as soon as its last gradient exists, so it
overlaps the rest of backward.

Works unchanged on the gloo backend with CPU tensors (that is how tests/test_ddp_gloo.py runs it).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


def rank() -> int:
    return dist.get_rank() if is_dist() else 0


class GradBuckets:
    """Flattens each bucket's gradients with one `cat`, all-reduces it asynchronously, and re-points
    `.grad` of every parameter at its slice of the reduced flat buffer (no copy back)."""

    def __init__(self, model: torch.nn.Module, process_group=None, overlap: bool = True):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if is_dist() else 1
        self.overlap = overlap
        groups: "OrderedDict[str, List[torch.nn.Parameter]]" = OrderedDict()
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            top = name.split(".")[1] if name.startswith("module.") else name.split(".")[0]
            groups.setdefault(top, []).append(p)
        # backward finishes decoders (last created first) before the encoder
        order = [k for k in reversed(list(groups.keys())) if k != "encoder"] + [k for k in groups if k == "encoder"]
        self.buckets: List[List[torch.nn.Parameter]] = [groups[k] for k in order]
        self.names = order
        self._pending = [0] * len(self.buckets)
        self._handles: List = []
        self._flat: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._hooks = []
        if self.world > 1 and overlap:
            for bi, params in enumerate(self.buckets):
                for p in params:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self.reset()

    def reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._handles = []

    def _make_hook(self, bi: int):
        def hook(_param):
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self._launch(bi)
        return hook

    def _launch(self, bi: int):
        params = self.buckets[bi]
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        self._flat[bi] = flat
        h = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._handles.append((bi, h))

    def finish(self):
        """Call after backward(): waits for the collectives, averages, re-points the gradients."""
        if self.world == 1:
            return
        if not self.overlap:
            for bi in range(len(self.buckets)):
                self._launch(bi)
        else:
            for bi, left in enumerate(self._pending):      # parameters that received no gradient
                if left != 0:
                    raise RuntimeError(f"bucket {self.names[bi]}: {left} parameters got no gradient this step")
        inv = 1.0 / self.world
        for bi, h in self._handles:
            h.wait()
            flat = self._flat[bi]
            flat.mul_(inv)
            off = 0
            for p in self.buckets[bi]:
                n = p.numel()
                p.grad = flat[off:off + n].view_as(p)
                off += n
        self.reset()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_model(model: torch.nn.Module, src: int = 0, group=None):
    """Same initial parameters and BatchNorm buffers on every rank (nn.DataParallel's `replicate`)."""
    if world_size() == 1:
        return
    with torch.no_grad():                      # in place on the parameter itself: bumps its version counter,
        for t in list(model.parameters()) + list(model.buffers()):   # which the packed-weight cache keys on
            dist.broadcast(t, src=src, group=group)
