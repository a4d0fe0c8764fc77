"""Who owns the host-side state of a training step (round 6).

A step's forward runs on the thread that calls the trainer; its backward runs on autograd's worker threads, which every backward of
the process shares.  State that both halves touch therefore cannot be a module global (two trainers stepping from two threads
would mix their entries) and cannot be thread-local either (the backward would not find what the forward left).  It lives in a
`StepContext`:

  * the trainer's scope (`conv.deferred_reduces`, which UAPSTrainer / BaselineTrainer / StepGraph open around forward + backward)
    creates one and makes it the CURRENT context of the calling thread;
  * every autograd Function whose backward needs it reads `current()` in its forward and keeps the reference on its ctx
    (`ctx.step`), so the backward -- whatever thread runs it -- finds the context of the step it belongs to;
  * the data-parallel bucket hooks get it from the trainer (`GradBuckets.step`).

What only the forward touches (the "track max|output| of the next convolution" request of UNet_UAPS, the statistics-group count,
the side stream the perturbed copies are written on, the depth of lazybn scopes, an optional private perturbation RNG) is per
thread: `fwd()`.

Nothing here is needed by a plain user loop (`model(x)`, `loss.backward()`, `optimizer.step()`, UAPS_train.py:285-292): without a
scope `current()` is None and every convolution reduces its weight gradient at once, every BatchNorm runs its one-piece backward.
"""
from __future__ import annotations

import threading
from typing import Callable, List, Optional


class StepContext:
    """One training step's forward + backward, as far as the host has to remember it."""
    __slots__ = ("deferred", "early", "early_hook", "early_done", "owner")

    def __init__(self, on_early: Optional[Callable] = None, owner=None):
        self.deferred: Optional[List] = []       # conv._Pending records: weight gradients whose partials are queued, reduction not yet launched
        self.early = None                        # (side stream, items) of the early flush until the final flush has joined the stream
        self.early_hook = on_early               # called on the side stream behind the early flush's launches (the trainer's early Adam)
        self.early_done = False
        self.owner = owner                       # the trainer, for error messages


class _Forward(threading.local):
    """Per-thread scratch of the forward pass (a model's forward runs on ONE thread from start to end)."""

    def __init__(self):
        self.step: Optional[StepContext] = None  # the open trainer scope of this thread
        self.amax_request = False                # conv.request_out_amax(): the next forward convolution raises a bound to max|its output|
        self.last_out_amax = None
        self.last_up2x_amax = None               # fused.upsample2x: the bound of the tensor it just wrote
        self.stat_groups = 1                     # fused.stat_groups(n)
        self.fan_side = None                     # perturb: the stream the perturbed feature copies are written on (None: the caller's)
        self.lazy_scope = None                   # lazybn.scope(): the innermost open scope object
        self.rng = None                          # perturb.local_rng(): this thread's private (seed, offset) stream, else the process's


_fwd = _Forward()


def fwd() -> _Forward:
    return _fwd


def current() -> Optional[StepContext]:
    """The step scope open on the calling thread (None outside a trainer's step)."""
    return _fwd.step


def push(ctx: StepContext) -> Optional[StepContext]:
    prev, _fwd.step = _fwd.step, ctx
    return prev


def pop(prev: Optional[StepContext]) -> None:
    _fwd.step = prev
