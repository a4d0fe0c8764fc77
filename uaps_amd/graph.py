"""Captured training step: the whole UAPS step (forward_pair, loss block, backward, Adam, confusion matrix) recorded once as a
hipGraph (torch.cuda.CUDAGraph) and replayed with ONE launch per step.

Why: a step is ~450 kernel launches driven by ~9 ms of Python (autograd Functions + ctypes); at 16 + 16 images the GPU work
(17-18 ms) hides that, at small batches it is the whole step.  A replay also lets the four decoder chains overlap on the GPU
(the capture records the per-decoder streams of UNet_UAPS as parallel branches) without any host involvement.

What changes per step can no longer travel in kernel arguments (a graph freezes them), so it lives in the device-resident
*step state* (include/uaps_hip.h, uaps_set_step_state): the Philox key increment, the Dirichlet mixing weights, the two
consistency weights and Adam's two step scalars.  The host fills a slot of a ring of pinned mirrors before every step and enqueues its
upload in front of the replay (stream-ordered).  `UAPSTrainer(step_state=True)` runs the same code path eagerly (bit-identical to the replay, which is
how tests/test_gpu_graph.py pins the capture); `use_graph=True` adds the capture.

Data parallel (world size > 1, standard gradient averaging): the step is captured as TWO graphs -- forward_pair + loss +
backward, whose gradients land in the flat bucket buffers of dist.GradBuckets, and Adam + confusion matrix -- with the
bucket all-reduces issued eagerly between the two replays: the collectives never enter a capture, every rank issues them
in the same order whatever it replays or runs eagerly, and only their overlap with the backward is given up (15 MB over
xGMI against a 14 ms step).  The gathered-loss mode (an exchange in the middle of the loss block) stays eager.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, bounds, conv, losses, metrics, perturb, lazybn

_KEY_STEP = 0x9E3779B97F4A7C15
NAN = float("nan")


class StepState:
    """Device buffer of the 16-word step state and a ring of pinned host mirrors.

    The upload is asynchronous and the host runs steps ahead of the GPU (a replay costs it half a millisecond), so a single
    pinned mirror would be overwritten for step n + 1 before the copy of step n has executed: every step fills its own ring
    slot, and a slot is reused only after the event recorded behind its last copy has completed.  The copy is issued on the
    step's stream in front of the replay -- it is not a node of the graph, whose source address would be fixed."""
    RING = 8

    def __init__(self, device: torch.device):
        self.hosts = [torch.zeros(16, dtype=torch.int32).pin_memory() for _ in range(self.RING)]
        self.events = [None] * self.RING
        self.slot = 0
        self.dev = torch.zeros(16, dtype=torch.int32, device=device)
        self.key = 0

    def fill(self, w, cw1: float, cw2: float, adam_step_size: float, adam_inv_sqrt_bc2: float) -> None:
        self.slot = (self.slot + 1) % self.RING
        if self.events[self.slot] is not None:
            self.events[self.slot].synchronize()          # RING steps ago: done long since
        host = self.hosts[self.slot].numpy()
        u32, f32 = host.view(np.uint32), host.view(np.float32)
        self.key = (self.key + _KEY_STEP) & 0xFFFFFFFFFFFFFFFF
        u32[0], u32[1] = self.key & 0xFFFFFFFF, self.key >> 32
        f32[2:10] = 0.0
        f32[2:2 + len(w)] = np.asarray(w, dtype=np.float64).astype(np.float32)
        f32[10], f32[11] = cw1, cw2
        f32[12], f32[13] = adam_step_size, adam_inv_sqrt_bc2

    def upload(self) -> None:
        """Enqueue the copy of the slot fill() wrote on the current stream (never inside a capture)."""
        self.dev.copy_(self.hosts[self.slot], non_blocking=True)
        ev = self.events[self.slot]
        if ev is None:
            ev = self.events[self.slot] = torch.cuda.Event()
        ev.record()

    def activate(self) -> None:
        _lib.check(_lib.lib().uaps_set_step_state(self.dev.data_ptr()), "uaps_set_step_state")

    @staticmethod
    def deactivate() -> None:
        _lib.check(_lib.lib().uaps_set_step_state(None), "uaps_set_step_state")


class StepGraph:
    """State-mode step of a UAPSTrainer, eager or captured.  Owned by the trainer (trainer.train_step dispatches here)."""

    def __init__(self, trainer, capture: bool, warmup: int = 2):
        self.split = trainer.buckets is not None        # data parallel: two graphs around the eager gradient exchange
        if self.split and trainer.gathered_loss:
            raise ValueError("the captured step does not cover the gathered-loss exchange (use the eager step)")
        if not trainer.pair_forward:
            raise ValueError("the captured step needs the forward_pair path (a GPU model)")
        self.tr = trainer
        self.state = StepState(trainer.device)
        self.want_capture, self.warmup = capture, max(1, warmup)     # Adam's lazily created moments must exist before a capture
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.graph_tail: Optional[torch.cuda.CUDAGraph] = None      # split form: Adam + confusion matrix
        self.calls = 0
        self.static = None
        self.live_outputs = False                # True: train_step returns the graph's own scalar tensors (overwritten by the next replay)
        self._side = torch.cuda.Stream(device=trainer.device) if capture else None
        # data parallel + capture: every step of every rank, warm-up and fallback steps included, launches the bucket
        # all-reduces in ONE fixed order after the backward (exchange_all), never from the backward hooks -- a rank whose
        # capture failed, or that steps eagerly for another reason, still pairs the same buckets as the ranks that replay
        self.fixed_order = self.split and capture
        trainer.optimizer.from_step_state = True

    def inputs(self):
        """The graph's own input tensors (x_l, y_l, x_u) once the step is captured, else None: a data pipeline that writes a batch
        straight into them (and passes them to train_step) spares the replay its three input copies."""
        s = self.static
        return None if s is None else (s["x_l"], s["y_l"], s["x_u"])

    def invalidate(self) -> None:
        """Forget the capture (checkpoint load, anything that replaced tensors the graph addresses): the next steps warm up
        eagerly and the step is recorded again."""
        self.graph = self.graph_tail = self.static = None
        self.calls = 0

    # ---- host side of one step: what the reference's loop computes on the host (UAPS_train.py:251, 279-280, 292) ----
    def _refresh(self):
        tr = self.tr
        opt = tr.optimizer
        # Adam's step count is read from the optimizer state every step (host tensors: no device sync), so that an eager
        # fallback step or a loaded checkpoint in between cannot leave a cached count behind
        st = next((opt.state[p]["step"] for g in opt.param_groups for p in g["params"] if p in opt.state and "step" in opt.state[p]), None)
        ss, isb = opt.step_scalars((int(st) if st is not None else 0) + 1)
        w = tr.mix_rng.dirichlet(np.ones(tr.n_heads), size=1)[0]
        cw1, cw2 = tr.consistency_weights()
        self.state.fill(w, cw1, cw2, ss, isb)
        return w, cw1, cw2

    def _head(self, x_l, y_l, x_u):
        """Forward, loss block and backward; every per-step scalar comes from the step state (w = None, cw = NaN)."""
        tr = self.tr
        perturb.rng().offset = 0                 # the key changes every step: the counters may restart (and must, for replays)
        with lazybn.scope(), conv.deferred_reduces(on_early=tr._early_adam(), model=tr.model) as step:
            if tr.buckets is not None:
                tr.buckets.step = step
            both = tr.model.forward_pair(x_l, x_u)
            out = losses.uaps_pair_loss(both, y_l, None, NAN, NAN)
            tr.optimizer.zero_grad(set_to_none=True)
            out.loss.backward(gradient=tr._unit_gradient(out.loss))
        return out, both

    def _tail(self, both, x_l, y_l):
        tr = self.tr
        tr.optimizer.step()
        return metrics.seg_confusion(both[0][: x_l.shape[0]], y_l) if tr.track_metrics else None

    def _body(self, x_l, y_l, x_u):
        """The device work of one step, eagerly (state mode, warm-up steps)."""
        if self.fixed_order:
            self.tr.buckets.defer = True
            try:
                out, both = self._head(x_l, y_l, x_u)
            finally:
                self.tr.buckets.defer = False
            if not self.tr.buckets.in_place():
                self.tr.buckets.gather_in()
            self.tr.buckets.exchange_all()
            self.tr.buckets.reset()
            return out, self._tail(both, x_l, y_l)
        out, both = self._head(x_l, y_l, x_u)
        if self.split:
            self.tr.buckets.finish()                 # overlapped with the backward by the bucket hooks
        return out, self._tail(both, x_l, y_l)

    def _after_replay(self):
        """Host bookkeeping of a replayed optimizer.step(): Adam's host-resident step counters, and the packed-weight cache
        (the replay re-packed into the graph's own buffers; eager code -- validate() -- must pack the new weights itself)."""
        opt = self.tr.optimizer
        steps = [opt.state[p]["step"] for g in opt.param_groups for p in g["params"] if p in opt.state]
        torch._foreach_add_(steps, 1)
        conv.optimizer_stepped(opt)

    def step(self, x_l, y_l, x_u) -> Dict[str, torch.Tensor]:
        tr = self.tr
        if not tr.model.training:
            tr.model.train()
        prev = (conv.REPACK_IN_PLACE, perturb.DEVICE_THRESHOLDS)
        conv.REPACK_IN_PLACE, perturb.DEVICE_THRESHOLDS = True, True
        self.state.activate()
        try:
            w, cw1, cw2 = self._refresh()
            if self.graph is not None and self._matches(x_l, y_l, x_u):
                self._copy_in(x_l, y_l, x_u)
                self.state.upload()                    # stream-ordered in front of the replay
                self.graph.replay()
                if self.split:
                    self._exchange()
                    self.graph_tail.replay()
                self._after_replay()
                out, cm = self.static["out"], self.static["cm"]
            elif self.want_capture and self.calls >= self.warmup and self.graph is None:
                try:
                    out, cm = self._capture(x_l, y_l, x_u)
                except Exception as e:                 # keep training: the eager state-mode step is the same arithmetic
                    if not self.split:
                        raise
                    import sys
                    print(f"uaps_amd.graph: capture of the data-parallel step failed ({type(e).__name__}: {e}); continuing eagerly", file=sys.stderr)
                    self.want_capture, self.graph, self.graph_tail = False, None, None
                    tr.optimizer.zero_grad(set_to_none=True)
                    self.state.upload()
                    out, cm = self._body(x_l, y_l, x_u)
            else:
                if self._side is not None:             # warm-up steps of a capture run on a side stream (torch's capture recipe)
                    self._side.wait_stream(torch.cuda.current_stream(tr.device))
                    with torch.cuda.stream(self._side):
                        self.state.upload()
                        out, cm = self._body(x_l, y_l, x_u)
                    torch.cuda.current_stream(tr.device).wait_stream(self._side)
                else:
                    self.state.upload()
                    out, cm = self._body(x_l, y_l, x_u)
        finally:
            conv.REPACK_IN_PLACE, perturb.DEVICE_THRESHOLDS = prev
            StepState.deactivate()
        self.calls += 1
        if cm is not None:
            tr._cms.append(cm.clone() if self.graph is not None else cm)
        tr.iter_num += 1
        # a replay overwrites the graph's static output tensors: hand out copies (as for the confusion matrix above), so that
        # scalars a caller collects over an epoch keep their own step's values -- unless the caller asked for the live tensors
        # (`live_outputs`: a loop that reads the loss only now and then; three small library launches fewer per step)
        sc = (lambda t: t.detach().clone()) if (self.graph is not None and not self.live_outputs) else (lambda t: t.detach())
        tr.last = {"loss": sc(out.loss), "sup": sc(out.sup), "unsup": sc(out.unsup), "cw1": cw1, "cw2": cw2, "w": w}
        return tr.last

    # ---- capture / replay ----
    def _matches(self, x_l, y_l, x_u) -> bool:
        s = self.static
        return s is not None and x_l.shape == s["x_l"].shape and y_l.shape == s["y_l"].shape and x_u.shape == s["x_u"].shape

    def _copy_in(self, x_l, y_l, x_u) -> None:
        s = self.static
        for name, t in (("x_l", x_l), ("y_l", y_l), ("x_u", x_u)):
            if t.data_ptr() != s[name].data_ptr():
                s[name].copy_(t, non_blocking=True)

    def _capture(self, x_l, y_l, x_u):
        tr = self.tr
        dev = tr.device
        self.static = {"x_l": x_l.clone(), "y_l": y_l.clone(), "x_u": x_u.clone()}
        conv.invalidate_packed_weights()             # the weight packing must be part of the captured step
        bounds.reset_pool()                          # ... and so must the zero fill of every max|.| scalar the kernels raise
        tr.optimizer.zero_grad(set_to_none=True)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        if not self.split:
            # thread-local error mode on every capture (round 6): runtime calls of OTHER threads (a DataLoader's pin thread, RCCL's
            # watchdog) do not invalidate it; quiet_gc keeps this thread's own finalisers out of it
            with _lib.quiet_gc(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                out, cm = self._body(self.static["x_l"], self.static["y_l"], self.static["x_u"])
        else:
            # thread_local: the process group's watchdog thread may query its events while this thread captures
            tr.buckets.defer = True                  # the backward hooks launch nothing inside the capture
            try:
                with _lib.quiet_gc(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                    out, both = self._head(self.static["x_l"], self.static["y_l"], self.static["x_u"])
                if not tr.buckets.in_place():
                    raise RuntimeError("a gradient of the captured backward was not written into its bucket slice")
                g2 = torch.cuda.CUDAGraph()
                with _lib.quiet_gc(), torch.cuda.graph(g2, pool=g.pool(), capture_error_mode="thread_local"):
                    cm = self._tail(both, self.static["x_l"], self.static["y_l"])
            finally:
                tr.buckets.defer = False
                tr.buckets.reset()
            self.graph_tail = g2
            self.static["both"] = both
        self.graph = g
        bounds.reset_pool()                          # eager code must not be handed scalars the replays re-zero
        # what the replays update and step() hands out: the output tensors WITHOUT their autograd graph.  Keeping `out` itself kept every
        # node of the captured backward alive -- among them the parameters' AccumulateGrad nodes, created on the capture's streams, which
        # later eager steps on other streams then met again (PyTorch's "AccumulateGrad node's stream does not match" warning in
        # bench.py's analysis pass); the activations it held belong to the graph's private pool either way
        self.static["out"] = type(out)(*[(t.detach() if torch.is_tensor(t) else t) for t in out]) if isinstance(out, tuple) else out
        self.static["cm"] = cm
        out = self.static["out"]
        # the graph writes the packed weights through raw pointers: hold the buffers, whatever the cache does later
        self.static["packed"] = [(e[3], e[4]) for e in conv._packed.values()]
        # the capture executed nothing (and its host-side Adam counter bump belongs to no step): undo that, then run the step
        opt = tr.optimizer
        steps = [opt.state[p]["step"] for gr in opt.param_groups for p in gr["params"] if p in opt.state]
        torch._foreach_add_(steps, -1)
        self.state.upload()
        g.replay()
        if self.split:
            self._exchange()
            self.graph_tail.replay()
        self._after_replay()
        return out, cm

    def _exchange(self):
        """The gradient exchange between the two replays: every bucket all-reduced in place (and averaged), on the step's stream."""
        self.tr.buckets.exchange_all()
