"""Data-parallel exchange steps of the UAPS step: one process per GPU, RCCL over xGMI.

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

import contextlib
from collections import OrderedDict
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import config


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


def rank() -> int:
    return dist.get_rank() if is_dist() else 0


def exchange_sums(group=None):
    """The `exchange` callable of losses.uaps_pair_loss for gathered-batch loss statistics: sums the tensor of raw loss sums
    over the ranks in place and returns the number of ranks (a second, latency-only collective of
    D (2C + 3) + ... doubles between the loss block's forward and its finalize)."""
    def _x(t: torch.Tensor) -> int:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return dist.get_world_size(group)
    return _x


class GradBuckets:
    """One flat gradient buffer per bucket, allocated once.  Every parameter's slice (16-byte aligned) is registered as the
    destination of its gradient (_graddest): the backward kernels write there, autograd adopts the view as `.grad`, and the
    bucket's asynchronous all-reduce runs on the flat buffer in place as soon as the bucket's last gradient exists -- no
    per-step concatenation, no copy back.  A gradient that arrives elsewhere (an op without destination support) is copied
    into its slice first, so the result never depends on which path produced it.

    average=True divides by the world size (standard data parallelism: every rank's loss is a mean over its own shard);
    average=False leaves the SUM, for losses already normalised by the global pixel count (losses._PairLoss exchange)."""

    def __init__(self, model: torch.nn.Module, process_group=None, overlap: bool = True, average: bool = True):
        from . import _graddest
        self.group = process_group
        self.world = dist.get_world_size(process_group) if is_dist() else 1
        self.overlap = overlap
        self.average = average
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
        self._ids = [{id(p) for p in params} for params in self.buckets]
        self._pending = [0] * len(self.buckets)
        self._handles: List = []
        self._flat: List[torch.Tensor] = []
        self._offsets: List[List[int]] = []
        for params in self.buckets:
            offs, off = [], 0
            for p in params:
                offs.append(off)
                off += (p.numel() + 3) // 4 * 4                      # 16-byte aligned slices (vector loads in the Adam kernel)
            flat = torch.zeros(off, dtype=params[0].dtype, device=params[0].device)
            self._flat.append(flat)
            self._offsets.append(offs)
            for p, o in zip(params, offs):
                _graddest.register(p, flat, o)
        self._hooks = []
        self.step = None                       # the trainer's open step scope (conv.deferred_reduces), set by the trainer for each step:
                                               # the hook that completes a bucket reduces that bucket's deferred weight gradients first
        # the per-bucket reductions + all-reduce on a side stream beside the rest of the backward (UAPS_SIDE_EXCHANGE=0: in the
        # backward's own stream, as in round 5)
        self.side_exchange = config.flag("UAPS_SIDE_EXCHANGE", True)
        self.defer = False
        self.muted = False                     # hooks do nothing (a caller that steps without any exchange: bench.py's no-exchange leg)
        if self.world > 1 and overlap:
            for bi, params in enumerate(self.buckets):
                for p in params:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self.reset()

    def reset(self):
        from . import _graddest
        self._pending = [len(b) for b in self.buckets]
        self._handles = []
        _graddest.new_backward()

    def _make_hook(self, bi: int):
        def hook(_param):
            if self.defer or self.muted:       # captured backward (graph.StepGraph): the exchange is launched by finish(), after the replay
                return
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self._launch(bi)
        hook._uaps_bucket = True               # conv.leaf_refs: this hook reads .grad only when it launches the exchange (the trainers know when)
        return hook

    def _view(self, bi: int, k: int) -> torch.Tensor:
        p = self.buckets[bi][k]
        o = self._offsets[bi][k]
        return self._flat[bi][o:o + p.numel()].view_as(p)

    def _launch(self, bi: int):
        from . import conv
        flat = self._flat[bi]
        # Round 6: a bucket whose gradients are complete while the backward still runs (the decoders' buckets: the encoder's backward
        # follows) does its reductions and its all-reduce on a SIDE stream, beside the rest of the backward -- what the early flush
        # does at N = 1 (conv.early_flush) -- instead of in the backward's own stream, where the 200-us batched reduction of the
        # decoders' 52 weight gradients would sit in front of the encoder's kernels.  finish() joins the collectives as before.
        side = None
        if self.side_exchange and self.step is not None and self.overlap and not self.defer and flat.is_cuda:
            side = conv.reduce_stream(flat.device)
            side.wait_stream(torch.cuda.current_stream(flat.device))
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            conv.flush_params(self.step, self._ids[bi], keep_for=side)   # this bucket's deferred weight gradients, one launch
            for k, p in enumerate(self.buckets[bi]):
                v = self._view(bi, k)
                if p.grad is None:
                    raise RuntimeError(f"bucket {self.names[bi]}: a parameter got no gradient this step")
                if p.grad.data_ptr() != v.data_ptr():                    # produced outside the flat buffer: copy in, re-point
                    v.copy_(p.grad)
                    p.grad = v
            h = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._handles.append((bi, h, side))

    def finish(self):
        """Call after backward(): waits for the collectives and (average=True) divides by the world size."""
        if self.world == 1:
            return
        if not self.overlap or self.defer:
            for bi in range(len(self.buckets)):
                self._launch(bi)
        else:
            for bi, left in enumerate(self._pending):      # parameters that received no gradient
                if left != 0:
                    raise RuntimeError(f"bucket {self.names[bi]}: {left} parameters got no gradient this step")
        inv = 1.0 / self.world
        for bi, h, side in self._handles:
            h.wait()
            if side is not None:                           # (a gloo collective completes on the host: the side stream's reductions and
                torch.cuda.current_stream(self._flat[bi].device).wait_stream(side)      # copies in front of it are joined explicitly)
            if self.average:
                self._flat[bi].mul_(inv)
        self.reset()

    def in_place(self) -> bool:
        """True when every parameter's .grad is its slice of the flat buffer (what a captured backward must guarantee: a gradient
        produced elsewhere would need the per-step copy of _launch, which a replay does not run)."""
        return all(p.grad is not None and p.grad.data_ptr() == self._view(bi, k).data_ptr()
                   for bi, params in enumerate(self.buckets) for k, p in enumerate(params))

    def gather_in(self):
        """Copy every gradient that was produced outside its slice into the flat buffers and re-point .grad (what _launch does
        per bucket)."""
        for bi, params in enumerate(self.buckets):
            for k, p in enumerate(params):
                v = self._view(bi, k)
                if p.grad is None:
                    raise RuntimeError(f"bucket {self.names[bi]}: a parameter got no gradient this step")
                if p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
                    p.grad = v

    def exchange_all(self):
        """All buckets at once, no hooks involved (between the two replays of graph.StepGraph): all-reduce, wait, average."""
        if self.world == 1:
            return
        hs = [dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for flat in self._flat]
        inv = 1.0 / self.world
        for flat, h in zip(self._flat, hs):
            h.wait()
            if self.average:
                flat.mul_(inv)

    def remove(self):
        from . import _graddest
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for params in self.buckets:
            for p in params:
                _graddest.unregister(p)


def broadcast_model(model: torch.nn.Module, src: int = 0, group=None):
    """Same initial parameters and BatchNorm buffers on every rank (nn.DataParallel's `replicate`)."""
    if world_size() == 1:
        return
    with torch.no_grad():                      # in place on the parameter itself: bumps its version counter,
        for t in list(model.parameters()) + list(model.buffers()):   # which the packed-weight cache keys on
            dist.broadcast(t, src=src, group=group)
