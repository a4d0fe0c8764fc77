"""Host side of the convolution kernels (csrc/conv_kernels.hpp, C ABI section "Convolutions").

`conv2d(x, weight, bias)` is nn.Conv2d's forward for the only two forms the reference U-Net uses
(3x3 / padding 1 and 1x1, stride 1: utilities/UAPS_unet.py:37,41,73,138) with a hand-written
backward: the input gradient is the same implicit-GEMM kernel on the transposed, tap-flipped packed
weights, the weight (and bias) gradient a pixel-split MFMA reduction.  There is no MIOpen / PyTorch
fallback on this path: CPU tensors are refused.
"""
from __future__ import annotations

import collections
import ctypes as C
import os
import weakref
from typing import Dict, Optional, Tuple

import torch
from torch.optim.optimizer import register_optimizer_step_post_hook

from . import _graddest, lazybn, _lib, bounds, config, stepctx

_ws: Dict[Tuple[int, int], torch.Tensor] = {}
# packed weights per parameter: id(weight) -> (weakref, version, generation, wf, wb)
_packed: Dict[int, tuple] = {}
# Fused / foreach optimizers (and this package's Adam kernel) update parameters without touching Tensor._version, so the version
# counter alone does not say when a packed copy or a BatchNorm parameter bound is stale.  What does: a counter of the OPTIMIZER that
# owns the parameter, bumped after each of its steps (global post-step hook) -- per optimizer, not per process (round 6): with one
# process-wide counter another trainer's step, from another thread, invalidated this model's bounds in the MIDDLE of its forward
# (bounds.bn_output_bound then returned None and that convolution silently ran the three-piece bf16 form: still fp32-accurate, but
# not the bits of the single-threaded run -- seen once as a failure of tests/test_gpu_threads.py inside the full suite).
_generation = 0            # bumped by invalidate_packed_weights(): manual invalidation of everything
_shared_cell = [0]         # the counter of parameters that more than one optimizer owns: every optimizer's step bumps it


def optimizer_stepped(optimizer, *_args, **_kwargs) -> None:
    """Post-step hook of every torch.optim optimizer (also called by graph.StepGraph behind a replayed step): the optimizer's
    parameters have changed.  Each parameter learns its optimizer's counter cell once."""
    cell = getattr(optimizer, "_uaps_cell", None)
    if cell is None:
        cell = optimizer._uaps_cell = [0]
    cell[0] += 1
    _shared_cell[0] += 1
    n = sum(len(g["params"]) for g in optimizer.param_groups)
    if getattr(optimizer, "_uaps_cell_n", -1) != n:          # first step, or add_param_group since
        for g in optimizer.param_groups:
            for p in g["params"]:
                other = getattr(p, "_uaps_cell", None)
                p._uaps_cell = cell if (other is None or other is cell) else _shared_cell
        optimizer._uaps_cell_n = n


def stamp(t: torch.Tensor) -> tuple:
    """What a cached derivative of parameter `t` (packed weights, a BatchNorm bound) is valid for."""
    c = getattr(t, "_uaps_cell", None)
    return (_generation, c[0] if c is not None else -1)


def invalidate_packed_weights(*_args, **_kwargs) -> None:
    """Forget every packed weight buffer and BatchNorm parameter bound.  Call it after writing to parameters through `.data` or other
    paths that bypass both the optimizers and the tensor version counter (optimizer steps are seen by themselves)."""
    global _generation
    _generation += 1


register_optimizer_step_post_hook(optimizer_stepped)


# bench.py sets KERNEL_EVENTS to a dict {kernel instantiation name: [(start_event, end_event, flops), ...]}
# to time individual launches with HIP events attached to the kernel's dispatch (_lib.LaunchTimer); EVENT_FILTER (a set of names)
# restricts the recording to those instantiations.  None (the default) records nothing.
KERNEL_EVENTS: Optional[Dict[str, list]] = None
EVENT_FILTER: Optional[set] = None
_variant_cache: Dict[tuple, str] = {}
ENOFORM = -4                      # UAPS_ENOFORM (include/uaps_hip.h)


def kernel_variant(kind: str, B: int, Cin: int, Cout: int, H: int, W: int, ks: int, cfg: int = 0) -> str:
    """Name of the kernel instantiation a call launches ('fwd' / 'bwd_data' / 'wrw'), as rocprofv3 prints it."""
    key = (kind, B, Cin, Cout, H, W, ks, cfg)
    name = _variant_cache.get(key)
    if name is None:
        buf = C.create_string_buffer(96)
        L = _lib.lib()
        if kind in ("wrw", "wrw_bn"):
            rc = L.uaps_conv_wrw_variant(B, Cin, Cout, H, W, ks, cfg, buf, 96)
        elif kind == "bwd_data":
            rc = L.uaps_conv_fwd_variant(B, Cout, Cin, H, W, ks, cfg, buf, 96)
        else:
            rc = L.uaps_conv_fwd_variant(B, Cin, Cout, H, W, ks, cfg, buf, 96)
        _lib.check(rc, "uaps_conv_*_variant")
        name = buf.value.decode()
        if kind.endswith("_bn"):      # the variants that apply BatchNorm + LeakyReLU while staging (fused._BnActConv)
            name = (name.replace("conv_fwd_kernel", "conv_fwd_bn_kernel").replace("conv_wrw_kernel", "conv_wrw_bn_kernel")
                    .replace("conv_sfwd_kernel", "conv_sfwd_bn_kernel").replace("conv_s32_kernel", "conv_s32_bn_kernel").replace("conv_s32t_kernel", "conv_s32t_bn_kernel").replace("conv_swrw_kernel", "conv_swrw_bn_kernel")
                    .replace("conv_small_kernel", "conv_small_bn_kernel").replace("conv_small_wrw_kernel", "conv_small_wrw_bn_kernel"))
        _variant_cache[key] = name
    return name


class _timed:
    def __init__(self, kind, B, Cin, Cout, H, W, ks, cfg, h16=False, stats=False, dt=False, name=None):
        self.on = False
        if KERNEL_EVENTS is not None and name is not None:      # the caller knows the instantiation (the up-sampling forms)
            self.name = name.replace("_kernel", "_dt_kernel", 1) if dt else name
            self.on = EVENT_FILTER is None or self.name in EVENT_FILTER
            self.flops = 2.0 * B * H * W * Cin * Cout * ks * ks
        elif KERNEL_EVENTS is not None:
            self.name = kernel_variant(kind, B, Cin, Cout, H, W, ks, cfg)
            if stats and self.name.startswith("conv_small"):      # the exact-N kernels have no statistics epilogue: the matrix kernel ran
                self.name = kernel_variant(kind, B, Cin, Cout, H, W, ks, cfg | (1 << 28))
            if h16:                   # every tensor operand carried a bound: the fp16 two-piece instantiation of the same plan ran
                kin = Cout if kind == "bwd_data" else Cin
                if (self.name.startswith("conv_small_") and "wrw" not in self.name and W % 256 == 0 and H % 16 == 0 and kin in (16, 24, 32)
                        and not (_lib.lib().uaps_conv_get_tuning() & (128 | 8))):
                    # <= 4 output channels on a 256-wide map: the full-width-row kernel (csrc/conv_fwd.hip: row16)
                    self.name = ("conv_hr16%s_bn_kernel" if "_bn_" in self.name else "conv_hr16%s_kernel") % ("w" if W > 256 else "") + ("<2>" if kin <= 16 else "<4>")
                if self.name.startswith("conv_sfwd") and "<3, 8, 32, 16," in self.name and 8 < kin <= 32:
                    # <= 16 output channels on a wide map: the persistent kernels of csrc/conv_split_n16.hpp
                    row = W % 256 == 0 and H % 16 == 0 and not (_lib.lib().uaps_conv_get_tuning() & 128)      # csrc/conv_fwd.hip: launch_hr16
                    fam = ("r16w" if W > 256 else "r16") if row else "p16"      # wider than 256: the column-strip instantiations
                    self.name = ("conv_h%s_bn_kernel" if "_bn_" in self.name else "conv_h%s_kernel") % fam + ("<2>" if kin <= 16 else "<4>")
                if (kind == "bwd_data" and ks == 3 and kin == 16 and Cin == 32 and W % 256 == 0 and H % 16 == 0 and self.name.startswith("conv_s32")
                        and not (_lib.lib().uaps_conv_get_tuning() & (128 | 8))):
                    self.name = "conv_hr16wx2_kernel" if W > 256 else "conv_hr16x2_kernel"       # 16 -> 16 + 16 channels: two output tiles of the full-width-row kernel
                kout = Cin if kind == "bwd_data" else Cout
                if (kind in ("fwd", "fwd_bn", "bwd_data") and ks == 3 and W == 32 and H % 4 == 0 and kin % 32 == 0 and kout % 128 == 0
                        and not (cfg & 0x7fffffff & ~BOUNDED) and (not stats or (cfg & BOUNDED))
                        and (self.name.startswith("conv_s32") or self.name.startswith("conv_sfwd"))
                        and not (_lib.lib().uaps_conv_get_tuning() & 2048)):
                    # 128-output-channel layers on 32-wide maps: the whole-layer-width tile form (csrc/conv_split_g.hpp; csrc/conv_fwd.hip: g128)
                    self.name = "conv_hg128_bn_kernel" if "_bn_" in self.name else "conv_hg128_kernel"
                self.name = (self.name.replace("conv_s32", "conv_h32").replace("conv_sfwd", "conv_hfwd").replace("conv_swrw", "conv_hwrw")
                             .replace("conv_g1s", "conv_g1h").replace("conv_gw1s", "conv_gw1h"))
                if self.name == "conv_g1h_kernel<128>" and (Cin if kind == "bwd_data" else Cout) % 256 == 0 and not (_lib.lib().uaps_conv_get_tuning() & 1024):
                    self.name = "conv_g1h256_kernel"       # 256 output channels per workgroup (csrc/conv_fwd.hip: launch_g1)
                if (kind in ("wrw", "wrw_bn") and ks == 3 and W % 256 == 0 and H % 16 == 0 and Cout <= 16 and Cin <= 32 and not (cfg >> 24)
                        and (self.name.startswith("conv_hwrw") or self.name.startswith("conv_small_wrw") or self.name.startswith("conv_wrw_"))
                        and not (_lib.lib().uaps_conv_get_tuning() & (256 | 2))):
                    # the full-width-row weight-gradient kernels (csrc/conv_split_wrw_row.hpp; csrc/conv_wrw.hip: the eligibility block)
                    self.name = ("conv_hrwrw%s_bn_kernel<%d>" if "_bn_" in self.name else "conv_hrwrw%s_kernel<%d>") % ("w" if W > 256 else "", 1 if Cin <= 16 else 2)
                if self.name.startswith("conv_hwrw") and "_kernel<4, 1, " in self.name and H >= 8 and not (_lib.lib().uaps_conv_get_tuning() & 32):
                    self.name = self.name.replace("_kernel<4, 1, ", "_kernel<8, 1, ")      # 16 output channels: the 8-row tiles (csrc/conv_wrw.hip: launch_swrw)
                if self.name.startswith("conv_wrw_kernel<3, 4, 32, 2, 2, 4, ") and not self.name.endswith(" 1>") and W >= 32 and Cin >= 16:
                    self.name = "conv_hwrw_d_kernel<4, 2, 2, " + self.name.split(",")[-1].strip()      # the dilated fp16 form (csrc/conv_wrw.hip)
            if dt:                    # the forms that turn d(activation) into dy while staging (uaps_call_hints::dyt_*)
                self.name = self.name.replace("_kernel", "_dt_kernel", 1)
            self.on = EVENT_FILTER is None or self.name in EVENT_FILTER
            self.flops = 2.0 * B * H * W * Cin * Cout * ks * ks

    def __enter__(self):
        self.armed = self.on
        if self.on:
            self.t = _lib.LaunchTimer().__enter__()
        return self

    def __exit__(self, *a):
        if self.armed:                # (a caller clears `on` when the entry point launched nothing: disarm, record nothing)
            self.t.__exit__()
            if self.on:
                KERNEL_EVENTS.setdefault(self.name, []).append((self.t.start, self.t.stop, self.flops))
        return False


def set_mode(mode) -> None:
    """Arithmetic of the convolution kernels: 'h16' / 2 (default) = two fp16 pieces per power-of-two-scaled operand where the
    operands' magnitude bounds are known (uaps_amd/bounds.py; 22 significant bits, fp32 accumulation), else as 'split';
    'split' / 1 = exact three-way bf16 split of both fp32 operands on the bf16 matrix pipe (csrc/conv_split.hpp, fp32-chain
    accuracy at 2.67x the fp32 matrix rate); 'exact' / 0 = the fp32 matrix instruction everywhere.  Process-wide; also
    UAPS_CONV_MODE=0/1/2 in the environment."""
    global _mode_name
    m = {"h16": 2, "split": 1, "bf16": 1, "exact": 0, "f32": 0}.get(mode, mode)
    _lib.check(_lib.lib().uaps_conv_set_mode(int(m)), "uaps_conv_set_mode")
    _variant_cache.clear()
    _parts_cache.clear()          # the plan (and with it the statistics layout of the GEMM-tiled 1x1 kernels) depends on the mode
    _mode_name = None


_mode_name = None      # cached (the hot path asks once per launch); uaps_conv_set_mode from elsewhere is not seen until set_mode()


def get_mode() -> str:
    global _mode_name
    if _mode_name is None:
        _mode_name = {2: "h16", 1: "split"}.get(_lib.lib().uaps_conv_get_mode(), "exact")
    return _mode_name


def _workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    key = (dev.index, _lib.current_stream(dev))
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8, device=dev)
        _ws[key] = w
    return w


# ---- deferred weight-gradient reductions ---------------------------------------------------------------------------------------
# A weight gradient is two launches: the matrix kernel that writes per-split partial sums and the fixed-order reduction of them,
# a 5-6 us launch that is all latency -- ~60 per training step.  Inside `deferred_reduces()` (the trainers open it around forward +
# backward of a step) every weight gradient keeps its partials in a buffer of its own and the reductions of the whole backward run
# as one launch per 28 gradients when the scope ends (`flush_weight_reduces`), bit-identical to the single launches.  Until then the
# gradient tensors autograd has been handed are allocated but NOT written: nothing may read a .grad inside the scope -- which is why
# it is a scope the step owns (like lazybn.scope) and not a global mode.  The one reader inside the backward this package has, the
# overlapped data-parallel exchange, reduces its bucket's gradients itself in front of the all-reduce (flush_params).
#
# Round 6: the scope's state is a stepctx.StepContext, not module globals.  The forward of every convolution Function keeps the
# context of the scope it ran in (`ctx.step`), so its backward -- on whichever autograd thread -- queues into the step it belongs to;
# two trainers stepping from two threads never see each other's entries (tests/test_gpu_threads.py).
_DEFER = config.flag("UAPS_DEFER_WRW_REDUCE", True)


def _foreign_grad_hooks(t: torch.Tensor) -> bool:
    """Does something this package does not know read t.grad INSIDE the backward?  Python-side tensor hooks and post-accumulate
    hooks are visible here; hooks that C++ code attaches to the parameter's AccumulateGrad node (torch.nn.parallel.
    DistributedDataParallel's reducer, FSDP) are NOT -- models wrapped that way are refused as a whole by `defer_allowed`."""
    if t._backward_hooks:
        return True
    post = getattr(t, "_post_accumulate_grad_hooks", None)
    return bool(post) and any(not getattr(h, "_uaps_bucket", False) for h in post.values())


def leaf_refs(weight: torch.Tensor, bias: Optional[torch.Tensor]):
    """Weak references to (weight, bias) when their gradients may be written late -- leaf tensors without tensor hooks and without
    foreign post-accumulate hooks (dist.GradBuckets' own are known), whose gradients go straight to AccumulateGrad (a non-leaf
    weight's gradient, e.g. conv3x3s2's re-arranged kernel, is read by the next autograd node at once) -- else None: that
    convolution reduces immediately whatever the scope."""
    for t in (weight, bias):
        if t is None:
            continue
        if not t.is_leaf or _foreign_grad_hooks(t):
            return None
    return (weakref.ref(weight), weakref.ref(bias) if bias is not None else None)


def defer_allowed(model) -> bool:
    """May a trainer open `deferred_reduces` for `model`?  Not for a model behind a gradient reducer this package cannot see:
    torch.nn.parallel.DistributedDataParallel and FSDP hang their hooks on the AccumulateGrad NODES (C++ side), read or copy the
    still unwritten .grad inside the backward, and nothing on the Python side of the parameter shows it (a DDP-wrapped
    model would train silently on unsynchronised gradients).  nn.DataParallel replicas and plain modules are fine; this package's own
    exchange (dist.GradBuckets) is handled through flush_params."""
    import torch.nn as nn
    refuse = [nn.parallel.DistributedDataParallel]
    try:
        from torch.distributed.fsdp import FullyShardedDataParallel
        refuse.append(FullyShardedDataParallel)
    except Exception:
        pass
    mods = model.modules() if isinstance(model, nn.Module) else ()
    return not any(isinstance(m, tuple(refuse)) for m in mods)


# one deferred reduction: the partials' buffer, the addresses (and storages: kept alive) of the gradient tensors, the call's dimensions,
# the stream the partials were launched on, leaf_refs of the parameters
_Pending = collections.namedtuple("_Pending", "ws dw dw_storage dev db db_storage dims stream prefs")


def step_of(prefs) -> Optional[stepctx.StepContext]:
    """The forward of a convolution Function: the step context its backward will queue its reduction into (None: reduce at once)."""
    return stepctx.current() if prefs is not None else None


def _defers(prefs, step) -> bool:
    return step is not None and step.deferred is not None and prefs is not None


def _wrw_workspace(dev: torch.device, nbytes: int, prefs=None, step=None) -> torch.Tensor:
    if _defers(prefs, step):
        return torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    return _workspace(dev, nbytes)


def _wrw_reduce(ws, dw, db, B, Cin, Cout, H, W, ks, cfg, st, prefs=None, step=None) -> None:
    """The reduction of one weight gradient's partials: now, or at the end of the deferred_reduces() scope `step` belongs to (prefs =
    leaf_refs(weight, bias) of the convolution, step = step_of(prefs) taken in the forward; same values as given to _wrw_workspace)."""
    if _defers(prefs, step):
        # addresses and storages, not the tensors: AccumulateGrad takes a gradient over as .grad only while nobody else holds it
        # (it copies otherwise -- here it would copy memory that is not written yet)
        step.deferred.append(_Pending(ws, dw.data_ptr(), dw.untyped_storage(), dw.device, db.data_ptr() if db is not None else None,
                                      db.untyped_storage() if db is not None else None, (B, Cin, Cout, H, W, ks, cfg),
                                      torch.cuda.current_stream(dw.device), prefs))
        return
    rc = _lib.lib().uaps_conv_bwd_weight_reduce(ws.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None, B, Cin, Cout,
                                                H, W, ks, cfg, st)
    _lib.check(rc, "uaps_conv_bwd_weight_reduce")


_reduce_streams: Dict[int, "torch.cuda.Stream"] = {}
_EARLY = config.flag("UAPS_EARLY_WRW_REDUCE", True)


def _launch_reduces(items: list, dev: torch.device) -> None:
    arr = (_lib.WrwReduceItem * len(items))()
    for a, it in zip(arr, items):
        a.workspace, a.dw, a.dbias = it.ws.data_ptr(), it.dw, it.db
        a.B, a.Cin, a.Cout, a.H, a.W, a.ks, a.cfg = it.dims
    with _lib.device_guard(dev):
        rc = _lib.lib().uaps_conv_bwd_weight_reduce_batch(arr, len(items), _lib.current_stream(dev))
    _lib.check(rc, "uaps_conv_bwd_weight_reduce_batch")


def early_flush(step: Optional[stepctx.StepContext]) -> int:
    """Called by the backward of a feature fan-out (perturb._FanOut, with the step context its forward saw): when the first of them
    runs -- the deepest feature's, whose gradient needs every decoder's whole backward -- all decoder weight gradients have their
    partials queued and the calling stream has been made to wait for the decoder streams by autograd.  Their reductions (most of the
    step's: 52 of 62 in the U-Net) go to a side stream here and run beside the encoder's backward, where the chip has one kernel at a
    time, instead of alone behind it; the scope's final flush joins the stream.  Once per scope; nothing outside a scope."""
    if step is None or not _EARLY or not step.deferred or step.early is not None or step.early_done:
        return 0
    from . import unet
    if not unet._DECODER_STREAMS:        # single-stream mode: every launch on the caller's stream
        return 0
    dev = step.deferred[0].dev
    if any(it.dev != dev for it in step.deferred):
        return 0
    side = reduce_stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    for st in {it.stream for it in step.deferred}:   # ... and for every stream partials were launched on (the caller's wait covers only
        side.wait_stream(st)                         # the producers of ITS inputs: true for all of them in UNet_UAPS, not in general)
    items = list(step.deferred)
    _verify(items, early=True)          # (before the hook below updates parameters from these gradients: ADVICE r5)
    with torch.cuda.stream(side):
        _launch_reduces(items, dev)
        if step.early_hook is not None:     # (UAPSTrainer: Adam for the parameters whose gradients are final now, behind their reductions)
            step.early_hook()
    step.early = (side, items)          # the partials' buffers stay alive until the final flush has joined the side stream
    step.early_done = True
    step.deferred.clear()
    return len(items)


def _verify(items, early: bool = False) -> None:
    """early: called from inside the backward (early_flush) -- a parameter whose AccumulateGrad has not run yet has no .grad; that is
    not an error there (the early hook leaves it to the final optimizer step), a .grad at ANOTHER address is."""
    for it in items:
        for ref, ptr in ((it.prefs[0], it.dw), (it.prefs[1], it.db)):
            t = ref() if (ref is not None and ptr is not None) else None
            if t is not None and t.requires_grad and ((t.grad is None and not early) or (t.grad is not None and t.grad.data_ptr() != ptr)):
                raise RuntimeError("deferred weight-gradient reduction: a parameter's .grad is not the buffer its reduction wrote (the "
                                   "unwritten gradient was summed or copied by autograd: shared weights, or gradient accumulation over "
                                   "several backward passes); set UAPS_DEFER_WRW_REDUCE=0 for this training loop")


def reduce_stream(dev: torch.device) -> "torch.cuda.Stream":
    """The side stream of `dev` that reductions taken off the backward's own stream run on (early_flush; dist.GradBuckets._launch)."""
    side = _reduce_streams.get(dev.index)
    if side is None:
        side = _reduce_streams[dev.index] = torch.cuda.Stream(device=dev)
    return side


def flush_params(step: Optional[stepctx.StepContext], param_ids, keep_for=None) -> int:
    """The pending reductions of the convolutions whose weight is one of `param_ids` (ids of parameters), on the current stream:
    dist.GradBuckets calls this from the gradient hook that completes a bucket, in front of the bucket's all-reduce -- the
    overlapped data-parallel exchange reads .grad inside the backward, so its reductions cannot wait for the end of the scope;
    one launch per bucket instead of one per convolution.  The number of gradients reduced.
    keep_for: the (side) stream this call runs on when that is not the stream the partials' buffers were allocated on -- they are
    recorded on it, so that the allocator does not hand them out again while the reduction is still queued there."""
    if step is None or not step.deferred:
        return 0
    mine, rest = [], []
    for it in step.deferred:
        w = it.prefs[0]()
        (mine if (w is not None and id(w) in param_ids) else rest).append(it)
    if not mine:
        return 0
    dev = mine[0].dev
    cur = torch.cuda.current_stream(dev)
    for st in {it.stream for it in mine}:
        if st != cur:
            cur.wait_stream(st)
    _launch_reduces(mine, dev)
    if keep_for is not None:
        for it in mine:
            it.ws.record_stream(keep_for)
    _verify(mine)
    step.deferred[:] = rest
    return len(mine)


def flush_weight_reduces(step: Optional[stepctx.StepContext] = None) -> int:
    """Run the pending reductions of `step` (default: the calling thread's open scope) on the current stream of each device (after
    the backward has returned, i.e. after autograd has joined its streams) and join the early flush's side stream; the number of
    gradients reduced in the scope.  Raises if a parameter's .grad is not the buffer its reduction wrote (autograd summed or copied the
    unwritten gradient: a weight used by two convolutions, gradient accumulation over several backward passes -- such training loops
    set UAPS_DEFER_WRW_REDUCE=0)."""
    if step is None:
        step = stepctx.current()
    if step is None:
        return 0
    early, step.early = step.early, None
    done = []
    if early is not None:
        side, done = early
        torch.cuda.current_stream(done[0].dev).wait_stream(side)
    if step.deferred:
        items, step.deferred = step.deferred, []
        by_dev: Dict[torch.device, list] = {}
        for it in items:
            by_dev.setdefault(it.dev, []).append(it)
        for dev, its in by_dev.items():
            _launch_reduces(its, dev)
        done = done + items
    _verify(done)
    return len(done)


class deferred_reduces:
    """Scope of a training step's forward + backward: see the comment above.  `with deferred_reduces(...) as step:` -- step is the
    StepContext (None when deferral is off).  The scope belongs to the thread that opens it; nested scopes join the outer one.  An
    exception inside drops the pending reductions (their gradients stay unwritten, like the step they belonged to).
    model: refused (the scope then defers nothing) when it is wrapped in a gradient reducer this package cannot see
    (`defer_allowed`: torch DistributedDataParallel / FSDP)."""

    def __init__(self, enabled: bool = True, on_early=None, model=None):
        self.enabled = bool(enabled) and _DEFER and (model is None or _allowed_cached(model))
        self.outer = False
        self.on_early = on_early
        self.step = None
        self.prev = None

    def __enter__(self):
        if self.enabled:
            cur = stepctx.current()
            self.outer = cur is None
            if self.outer:
                self.step = stepctx.StepContext(self.on_early)
                self.prev = stepctx.push(self.step)
            else:
                self.step = cur
        return self.step

    def __exit__(self, exc_type, exc, tb):
        if self.enabled and self.outer:
            step = self.step
            try:
                if exc_type is None:
                    flush_weight_reduces(step)
                elif step.early is not None:          # a failed step: the side stream's launches are joined, nothing else is reduced
                    torch.cuda.current_stream(step.early[1][0].dev).wait_stream(step.early[0])
            finally:
                step.deferred, step.early, step.early_hook = None, None, None
                stepctx.pop(self.prev)
        return False


_allowed: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def _allowed_cached(model) -> bool:
    try:
        ok = _allowed.get(model)
        if ok is None:
            ok = _allowed[model] = defer_allowed(model)
        return ok
    except TypeError:                                # (not weak-referenceable: decide every time)
        return defer_allowed(model)


def _remember(weight: torch.Tensor, wf, wb) -> None:
    """Cache entry of `weight`, dropped the moment the tensor dies (a transient weight -- conv3x3s2's re-arranged kernel is a fresh
    non-leaf tensor every forward -- must not keep its packed buffers until some later purge)."""
    key = id(weight)

    def _gone(r, key=key, _packed=_packed):
        ent = _packed.get(key)
        if ent is not None and ent[0] is r:          # not an entry of a newer tensor that got the same id
            del _packed[key]
    _packed[key] = (weakref.ref(weight, _gone), weight._version, stamp(weight), wf, wb)


def pack_weights(weight: torch.Tensor, need_bwd: bool = True):
    """Packed forward / input-gradient weight buffers of `weight` [Cout,Cin,ks,ks], cached until the
    parameter is modified in place (optimizer step) or replaced."""
    key = id(weight)
    ent = _packed.get(key)
    if ent is not None:
        ref, ver, gen, wf, wb = ent
        if (ref() is weight and ver == weight._version and gen == stamp(weight) and wf.device == weight.device
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
    with _lib.device_guard(dev):
        rc = L.uaps_conv_pack_weights(w.data_ptr(), Cout, Cin, ks, wf.data_ptr(), wb.data_ptr() if need_bwd else None,
                                      _lib.current_stream(dev))
    _lib.check(rc, "uaps_conv_pack_weights")
    _remember(weight, wf, wb)
    return wf, wb


_parts_cache: Dict[tuple, int] = {}

# Packed weights are written into FRESH buffers whenever a parameter changed: the packed input-gradient weights `wb` are what
# the conv Functions save for backward, and a forward -> optimizer.step() -> forward -> backward(first graph) sequence must
# still see the weights of its own forward (the kernels write through raw pointers, so autograd's version counters cannot
# flag an in-place re-pack).  A captured hipGraph needs stable addresses instead: trainer graph capture sets this True.
# The cache is stream-ordered with the stream that packed (the step's main stream).
REPACK_IN_PLACE = False


def stats_parts_per_image(B: int, Cin: int, Cout: int, H: int, W: int, ks: int, cfg: int = 0) -> int:
    key = (B, Cin, Cout, H, W, ks, cfg)
    n = _parts_cache.get(key)
    if n is None:
        out = C.c_int()
        _lib.check(_lib.lib().uaps_conv_fwd_stats_parts(B, Cin, Cout, H, W, ks, cfg, C.byref(out)), "uaps_conv_fwd_stats_parts")
        n = _parts_cache[key] = out.value
    return n


_pack_streams: Dict[int, "torch.cuda.Stream"] = {}


def pack_all_beside(weights, dev: torch.device):
    """pack_all(weights) on a side stream that starts behind the current one: the caller goes on with work that does not read these
    weights (UNet_UAPS: the encoder's forward, while the decoders' weights -- 70 % of the step's packing -- are packed) and makes the
    consumer wait for the returned stream (None: nothing was stale, nothing launched)."""
    if pack_all(weights, dry=True) == 0:
        return None
    side = _pack_streams.get(dev.index)
    if side is None:
        side = _pack_streams[dev.index] = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        pack_all(weights)
    return side


def pack_all(weights, dry: bool = False) -> int:
    """Pack every stale weight of `weights` (conv parameters [Cout,Cin,k,k] on one device) with ONE kernel launch
    per 64 convolutions instead of one launch each; pack_weights() then finds them in the cache.  UNet_UAPS calls this
    at the start of a forward, i.e. once per optimizer step.  Returns the number of stale weights (dry: counts only)."""
    todo = []
    for wt in weights:
        ent = _packed.get(id(wt))
        if ent is not None and ent[0]() is wt and ent[1] == wt._version and ent[2] == stamp(wt) and ent[3].device == wt.device \
                and ent[4] is not None:
            continue
        todo.append(wt)
    if not todo or dry:
        return len(todo)
    L = _lib.lib()
    dev = todo[0].device
    n = len(todo)
    bufs = []
    for wt in todo:
        Cout, Cin, ks, ks2 = wt.shape
        if ks != ks2 or ks not in (1, 3) or wt.device != dev or wt.dtype != torch.float32:
            raise ValueError("pack_all: fp32 3x3 / 1x1 conv weights on one device expected")
        ent = _packed.get(id(wt))
        if REPACK_IN_PLACE and ent is not None and ent[0]() is wt and ent[3].device == dev and ent[4] is not None:
            wf, wb = ent[3], ent[4]                  # same parameter, new values, same addresses (captured graphs replay them)
        else:
            nf, nb = C.c_size_t(), C.c_size_t()
            _lib.check(L.uaps_conv_pack_floats(Cout, Cin, ks, C.byref(nf), C.byref(nb)), "uaps_conv_pack_floats")
            wf = torch.empty(nf.value, dtype=torch.float32, device=dev)
            wb = torch.empty(nb.value, dtype=torch.float32, device=dev)
        bufs.append((wt.detach().contiguous(), wf, wb))
    arr = lambda vals: (C.c_void_p * n)(*vals)
    ints = lambda vals: (C.c_int * n)(*vals)
    with _lib.device_guard(dev):
        rc = L.uaps_conv_pack_weights_batch(arr([b[0].data_ptr() for b in bufs]), arr([b[1].data_ptr() for b in bufs]),
                                            arr([b[2].data_ptr() for b in bufs]), ints([t.shape[0] for t in todo]),
                                            ints([t.shape[1] for t in todo]), ints([t.shape[2] for t in todo]), n,
                                            _lib.current_stream(dev))
    _lib.check(rc, "uaps_conv_pack_weights_batch")
    for wt, (_, wf, wb) in zip(todo, bufs):
        _remember(wt, wf, wb)
    return len(todo)


def _h16(*bs) -> bool:
    return all(b is not None for b in bs) and get_mode() == "h16"


_EXACT = 1 << 28          # cfg bit 28: the exact fp32 kernels (include/uaps_hip.h)
BOUNDED = 1 << 11         # UAPS_CONV_BOUNDED: every tensor operand of this forward call carries a bound (the planner may then pick a kernel
                          # of the fp16-split arithmetic whose BatchNorm-partials layout differs: csrc/conv_split_g.hpp)


def stats_cfg(cfg: int, *bs) -> int:
    """cfg of a forward call WITH BatchNorm partial sums: + UAPS_CONV_BOUNDED when all operand bounds `bs` are present in mode 'h16'.
    The same value goes to stats_parts_per_image and to the launch (they must agree on the partial-sum layout)."""
    return cfg | BOUNDED if _h16(*bs) else cfg


def plan_cfg(ks: int, cfg: int, plain: bool, *tensors) -> int:
    """The ONE place that decides whether a 1x1 convolution may take the GEMM-tiled plan (csrc/conv_gemm1x1.hpp: `g1`): that plan
    has its own BatchNorm-statistics and weight-gradient workspace layouts and only a single-tensor, 16-byte-aligned form, so a
    two-tensor / BatchNorm-in-staging call (`plain` False) or an odd pointer gets cfg bit 28 (exact fp32 kernels) -- handed to
    every entry point of the layer (statistics parts, forward, both gradients, workspace, reduce), which therefore agree."""
    if ks == 1 and (not plain or any(t is not None and t.data_ptr() % 16 for t in tensors)):
        return cfg | _EXACT
    return cfg


# ---- the up-sampled half of an UpBlock's concatenation formed in the consuming kernels' staging (round 5) ----------------------
X2_UP2 = 1 << 10                  # UAPS_CONV_X2_UP2 (include/uaps_hip.h)
_FUSED_UP2 = config.flag("UAPS_FUSED_UP2", True)
# "the next forward convolution of this thread's model code raises a bound to max|its output|": per THREAD (stepctx.fwd()), a
# forward runs on one thread from the request to the take


def request_out_amax() -> None:
    """The next conv2d / bn_act_conv forward tracks max|output| (uaps_call_hints::out_amax; the fp32-instruction kernels have it):
    the 1x1 projection in front of an up-sampling has no BatchNorm behind it to bound its output.  Fetch it with take_out_amax()."""
    f = stepctx.fwd()
    f.amax_request, f.last_out_amax = True, None


def _claim_amax(dev):
    f = stepctx.fwd()
    if not f.amax_request:
        return None
    f.amax_request = False
    return bounds.new_amax(dev) if bounds.enabled() else None


def _set_out_amax(am) -> None:
    stepctx.fwd().last_out_amax = am


def take_out_amax():
    f = stepctx.fwd()
    am, f.last_out_amax, f.amax_request = f.last_out_amax, None, False
    return am


def up2_eligible(skip: torch.Tensor, weight: torch.Tensor) -> bool:
    """May conv2d_cat(skip, low, weight, up2=True) run (csrc/conv_fwd.hip / conv_wrw.hip: the UP2 forms)?  up4's first convolution
    at the metric's size: 16 + 16 -> <= 16 channels, 3x3, a 256-wide map with H % 16 == 0, fp16-split arithmetic, bounded skip."""
    if not (_FUSED_UP2 and skip.is_cuda and get_mode() == "h16" and bounds.get(skip) is not None):
        return False
    Cout, Cin, ks, _ = weight.shape
    B, C1, H, W = skip.shape
    return (ks == 3 and Cin == 32 and C1 == 16 and Cout <= 16 and W == 256 and H % 16 == 0 and skip.dtype == torch.float32
            and not (_lib.lib().uaps_conv_get_tuning() & (1 | 2 | 8 | 128 | 256)))


def conv_fwd_raw(x: torch.Tensor, wf: torch.Tensor, bias: Optional[torch.Tensor], Cout: int, ks: int, cfg: int = 0,
                 want_stats: bool = False, xb=None, stat_shift=None):
    """y = conv(x); with want_stats also the per-tile (sum, sum of squares) of y as float2 [Cout][B][parts_per_image]
    (the first pass of the BatchNorm that follows), returned as (y, stats, parts_per_image)."""
    B, Cin, H, W = x.shape
    y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    bp = bias.data_ptr() if bias is not None else None
    cfg = plan_cfg(ks, cfg, True, x, y)
    if want_stats:
        cfg = stats_cfg(cfg, xb)
        ppi = stats_parts_per_image(B, Cin, Cout, H, W, ks, cfg)
        stats = torch.empty((Cout, B, ppi, 2), dtype=torch.float32, device=x.device)
    am = _claim_amax(x.device)
    for attempt in range(2):
        with _lib.device_guard(x.device), _timed("fwd", B, Cin, Cout, H, W, ks, cfg, _h16(xb), want_stats) as tm:
            hh = None
            if xb is not None or (want_stats and stat_shift is not None) or am is not None:
                hh = _lib.mk_hints((xb,) if xb is not None else (), am, stat_shift if want_stats else None)
            if want_stats:
                rc = L.uaps_conv_fwd_stats_h(hh, x.data_ptr(), wf.data_ptr(), bp, y.data_ptr(), stats.data_ptr(), B, Cin, Cout, H, W, ks, cfg,
                                             _lib.current_stream(x.device))
            else:
                rc = L.uaps_conv_fwd_h(hh, x.data_ptr(), wf.data_ptr(), bp, y.data_ptr(), B, Cin, Cout, H, W, ks, cfg,
                                       _lib.current_stream(x.device))
            if rc == ENOFORM:
                tm.on = False
        if rc != ENOFORM or am is None:
            break
        am = None                                     # this layer's kernel cannot track max|y|: run it without (the caller falls back)
    _lib.check(rc, "uaps_conv_fwd")
    _set_out_amax(am)
    return (y, stats, ppi) if want_stats else y


# BatchNorm-backward sums in the input-gradient kernel's epilogue (uaps_call_hints::bsum_*): built and measured in round 5 -- the
# sums pass it removes (46 us at 5.8 TB/s) costs the row kernel 38 us (it moves bytes at 4.0 TB/s): a wash, so OFF unless asked for
# (profiles/r05_bn_sums_epilogue_ab.txt)
_FUSED_BSUM = config.flag("UAPS_FUSED_BN_SUMS", False)


def conv_bwd_data_raw(dy: torch.Tensor, wb: torch.Tensor, Cin: int, ks: int, cfg: int = 0, dyb=None, bsum=None):
    """bsum = (y, mean, invstd, gamma, beta, slope, groups): dx is d(activation) of the train-mode BatchNorm + LeakyReLU whose raw
    input is y; where the layer's kernel can (uaps_call_hints::bsum_*), its epilogue also forms that BatchNorm's backward sums --
    then (dx, partials, maxes) comes back for lazybn.prepare_from_partials, else (dx, None, None)."""
    B, Cout, H, W = dy.shape
    dev = dy.device
    dx = torch.empty((B, Cin, H, W), dtype=torch.float32, device=dev)
    cfg = plan_cfg(ks, cfg, True, dy, dx)
    L = _lib.lib()
    partials = maxes = None
    if bsum is not None and _FUSED_BSUM and dyb is not None and ks == 3 and Cin == 16 and Cout == 16 and W == 256 and H % 16 == 0:
        ppi = H // 16                              # one part per 16-row run of the full-width-row kernel
        partials = torch.empty((Cin, B, ppi, 2), dtype=torch.float32, device=dev)
        maxes = torch.empty(2 * bounds.FLOATS, dtype=torch.float32, device=dev)
        with _lib.device_guard(dev):
            _lib.check(L.uaps_zero_bounds(maxes.data_ptr(), maxes.numel(), _lib.current_stream(dev)), "uaps_zero_bounds")
            with _timed("bwd_data", B, Cin, Cout, H, W, ks, cfg, True, name="conv_hr16_bs_kernel") as tm:
                rc = L.uaps_conv_bwd_data_h(_lib.mk_hints((dyb,), bsum=(bsum[0], bsum[1], bsum[2], bsum[3], bsum[4], partials, maxes, bsum[5], bsum[6])),
                                            dy.data_ptr(), wb.data_ptr(), dx.data_ptr(), B, Cin, Cout, H, W, ks, cfg, _lib.current_stream(dev))
                if rc == ENOFORM:
                    tm.on = False
        if rc != ENOFORM:
            _lib.check(rc, "uaps_conv_bwd_data")
            return dx, partials, maxes
        partials = maxes = None
    with _lib.device_guard(dev), _timed("bwd_data", B, Cin, Cout, H, W, ks, cfg, _h16(dyb)):
        rc = L.uaps_conv_bwd_data_h(_lib.mk_hints((dyb,)) if dyb is not None else None,
                                    dy.data_ptr(), wb.data_ptr(), dx.data_ptr(), B, Cin, Cout, H, W, ks, cfg, _lib.current_stream(dev))
    _lib.check(rc, "uaps_conv_bwd_data")
    return (dx, None, None) if bsum is not None else dx


def conv_bwd_weight_raw(dy: torch.Tensor, x: torch.Tensor, ks: int, want_bias: bool, cfg: int = 0, wkey=None, bkey=None,
                        dyb=None, xb=None, lz=None, prefs=None, step=None):
    """wkey / bkey: id() of the weight / bias parameter, for a registered gradient destination (_graddest).
    lz (lazybn.Lazy): `dy` is d(activation) behind the BatchNorm that follows this convolution; the kernel forms the true dy while
    staging and writes it through -- returned as a third value (with its bound) -- or the stand-alone pass does, where the layer's
    kernel has no such form."""
    B, Cout, H, W = dy.shape
    Cin = x.shape[1]
    dev = dy.device
    L = _lib.lib()
    n = C.c_size_t()
    cfg = plan_cfg(ks, cfg, True, dy, x)
    _lib.check(L.uaps_conv_wrw_workspace_bytes(B, Cin, Cout, H, W, ks, cfg, C.byref(n)), "uaps_conv_wrw_workspace_bytes")
    ws = _wrw_workspace(dev, n.value, prefs, step)
    dw = _graddest.take(wkey, (Cout, Cin, ks, ks), dev)
    db = _graddest.take(bkey, (Cout,), dev) if want_bias else None
    dy_true = None
    with _lib.device_guard(dev):
        st = _lib.current_stream(dev)
        rc = lazybn.ENOFORM
        if lz is not None and xb is not None:
            dy_true = torch.empty_like(dy)
            with _timed("wrw", B, Cin, Cout, H, W, ks, cfg, True, dt=True) as tm:
                rc = L.uaps_conv_bwd_weight_partial_h(_lib.mk_hints((lz.bound, xb), dyt=(lz.y, lz.coef, dy_true, lz.slope, lz.groups)),
                                                      dy.data_ptr(), x.data_ptr(), int(want_bias), B, Cin, Cout, H, W, ks, cfg,
                                                      ws.data_ptr(), ws.numel(), st)
                tm.on = tm.on and rc != lazybn.ENOFORM
            if rc != lazybn.ENOFORM:
                bounds.put(dy_true, *lz.bound)
        if lz is not None and rc == lazybn.ENOFORM:
            dy_true = dy = lazybn.materialize(dy, lz)
            dyb = bounds.get(dy)
        if lz is None or rc == lazybn.ENOFORM:
            with _timed("wrw", B, Cin, Cout, H, W, ks, cfg, _h16(dyb, xb)):
                rc = L.uaps_conv_bwd_weight_partial_h(_lib.mk_hints((dyb, xb)) if (dyb is not None and xb is not None) else None,
                                                      dy.data_ptr(), x.data_ptr(), int(want_bias), B, Cin, Cout, H, W, ks, cfg,
                                                      ws.data_ptr(), ws.numel(), st)
        _lib.check(rc, "uaps_conv_bwd_weight_partial")
        _wrw_reduce(ws, dw, db, B, Cin, Cout, H, W, ks, cfg, st, prefs, step)
    if lz is not None:
        return dw, db, dy_true
    return dw, db


def _dil_cfg(dilation: int) -> int:
    if dilation not in (1, 2, 4):
        raise ValueError("conv2d: dilation 1, 2 or 4")
    return 0 if dilation == 1 else dilation << 24


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, want_stats=False, dilation=1, stat_shift=None):
        _lib.require_device(x, "conv2d")
        ctx.set_materialize_grads(False)          # no zero tensors for the unused gradient of `stats`
        if x.dtype != torch.float32 or weight.dtype != torch.float32:
            raise TypeError("conv2d: fp32 only (the reference trains in fp32)")
        x = x.contiguous()
        Cout, Cin, ks, _ = weight.shape
        if x.shape[1] != Cin:
            raise ValueError(f"conv2d: input has {x.shape[1]} channels, weight expects {Cin}")
        need_bwd = ctx.needs_input_grad[0]
        wf, wb = pack_weights(weight, need_bwd=True)
        cfg = _dil_cfg(dilation)
        ctx.save_for_backward(x, wb)
        ctx.meta = (Cin, Cout, ks, bias is not None, cfg)
        ctx.keys = (id(weight), id(bias) if bias is not None else None)
        ctx.prefs = leaf_refs(weight, bias)
        ctx.step = step_of(ctx.prefs)             # the trainer scope this forward ran in: the backward queues its reduction there
        ctx.xb = xb = bounds.get(x)
        if want_stats:
            y, stats, _ppi = conv_fwd_raw(x, wf, bias, Cout, ks, cfg, want_stats=True, xb=xb, stat_shift=stat_shift)
            stats._uaps_shifted = stat_shift is not None
            ctx.mark_non_differentiable(stats)
            return y, stats
        return conv_fwd_raw(x, wf, bias, Cout, ks, cfg, xb=xb)

    @staticmethod
    def backward(ctx, dy, *_unused):
        if dy is None:
            return None, None, None, None, None, None
        x, wb = ctx.saved_tensors
        Cin, Cout, ks, has_bias, cfg = ctx.meta
        lz = lazybn.take(dy)                   # dy is d(activation) behind the BatchNorm that follows: the weight gradient runs first
        dyb = bounds.get(dy)
        dy = dy.contiguous()
        dw = db = None
        want_w = ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])
        if lz is not None and not want_w:       # frozen weight and bias: no weight-gradient launch, no bucket slice taken
            dy, lz = lazybn.materialize(dy, lz), None
            dyb = bounds.get(dy)
        if lz is not None:
            dw, db, dy = conv_bwd_weight_raw(dy, x, ks, has_bias and ctx.needs_input_grad[2], cfg, *ctx.keys, xb=ctx.xb, lz=lz, prefs=ctx.prefs, step=ctx.step)
            dyb = bounds.get(dy)
        dx = conv_bwd_data_raw(dy, wb, Cin, ks, cfg, dyb=dyb) if ctx.needs_input_grad[0] else None
        if lz is None and want_w:
            dw, db = conv_bwd_weight_raw(dy, x, ks, has_bias and ctx.needs_input_grad[2], cfg, *ctx.keys, dyb=dyb, xb=ctx.xb, prefs=ctx.prefs, step=ctx.step)
        return dx, dw, db, None, None, None


def conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, dilation: int = 1) -> torch.Tensor:
    """F.conv2d(x, weight, bias, stride=1, padding=dilation * (k // 2), dilation=dilation) for 3x3 (dilation 1, 2, 4) and
    1x1 kernels."""
    return _Conv2d.apply(x, weight, bias, False, dilation)


def _mark(res, shifted, weight):
    # the statistics tensor remembers whether its sums are taken about a shift (fused.bn_act / bn_act_conv must tell the finalize);
    # the raw output is marked as one whose gradient may arrive with a pending BatchNorm transform (lazybn)
    res[1]._uaps_shifted = shifted
    lazybn.mark(res[0], weight)
    return res


def conv2d_with_stats(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, dilation: int = 1,
                      stat_shift=None):
    """conv2d that also returns the per-tile (sum, sum of squares) of its output, float32 [Cout, B, parts, 2], for
    fused.bn_act(..., stats=...): the BatchNorm statistics pass rides in the convolution's epilogue.
    stat_shift = (bn.running_mean or None, the conv bias that BatchNorm will add or None): the sums are then taken about
    running_mean - bias per channel (no cancellation in the variance of channels with |mean| >> std); hand the result to
    bn_act / bn_act_conv of that same BatchNorm."""
    return _mark(_Conv2d.apply(x, weight, bias, True, dilation, stat_shift), stat_shift is not None, weight)


class _Conv2dCat(torch.autograd.Function):
    """conv2d(torch.cat([x1, x2], dim=1), weight) without materialising the concatenation (UpBlock.forward,
    UAPS_unet.py:84-85): the kernels read the two tensors, the input gradient comes back as two tensors."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, want_stats, stat_shift=None, up2=False):
        """up2: x2 is the LOW-resolution tensor [B, C2, H/2, W/2]; the kernels up-sample it x2 (bilinear, align_corners) while they
        stage it (UAPS_CONV_X2_UP2) -- `self.up(x1)` of UAPS_unet.py:74-75 is never materialised.  Only where up2_eligible()."""
        _lib.require_device(x1, "conv2d_cat")
        ctx.set_materialize_grads(False)
        x1, x2 = x1.contiguous(), x2.contiguous()
        B, C1, H, W = x1.shape
        C2 = x2.shape[1]
        Cout, Cin, ks, _ = weight.shape
        ctx.up2 = bool(up2)
        if x2.shape != ((B, C2, H // 2, W // 2) if up2 else (B, C2, H, W)) or C1 + C2 != Cin:
            raise ValueError(f"conv2d_cat: inputs {tuple(x1.shape)} + {tuple(x2.shape)} do not concatenate to {Cin} channels")
        if C1 % 16:
            raise ValueError("conv2d_cat: the first tensor must have a multiple of 16 channels")
        wf, wb = pack_weights(weight, need_bwd=True)
        dev = x1.device
        cfg = plan_cfg(ks, 0, False)
        y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=dev)
        stats = None
        b1, b2 = bounds.get(x1), bounds.get(x2)
        ctx.xb = (b1, b2)
        fcfg = stats_cfg(cfg, b1, b2) if want_stats else cfg       # (the forward launch and its statistics layout only: ctx.meta keeps cfg)
        if want_stats:
            stats = torch.empty((Cout, B, stats_parts_per_image(B, Cin, Cout, H, W, ks, fcfg), 2), dtype=torch.float32, device=dev)
        ucfg = fcfg | (X2_UP2 if up2 else 0)
        with _lib.device_guard(dev), _timed("fwd", B, Cin, Cout, H, W, ks, fcfg, _h16(b1, b2), want_stats, name="conv_hr16_up_kernel" if up2 else None):
            hh = None
            if (b1 is not None and b2 is not None) or (want_stats and stat_shift is not None):
                hh = _lib.mk_hints((b1, b2) if (b1 is not None and b2 is not None) else (), None, stat_shift if want_stats else None)
            rc = _lib.lib().uaps_conv_fwd_cat_h(hh, x1.data_ptr(), C1, x2.data_ptr(), C2, wf.data_ptr(),
                                                bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                                stats.data_ptr() if want_stats else None, B, Cout, H, W, ks, ucfg, _lib.current_stream(dev))
        _lib.check(rc, "uaps_conv_fwd_cat")
        ctx.save_for_backward(x1, x2, wb)
        ctx.meta = (C1, C2, Cout, ks, bias is not None, cfg)
        ctx.keys = (id(weight), id(bias) if bias is not None else None)
        ctx.prefs = leaf_refs(weight, bias)
        ctx.step = step_of(ctx.prefs)
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y

    @staticmethod
    def backward(ctx, dy, *_unused):
        if dy is None:
            return None, None, None, None, None, None, None
        x1, x2, wb = ctx.saved_tensors
        C1, C2, Cout, ks, has_bias, cfg = ctx.meta
        up2 = ctx.up2
        ucfg = cfg | (X2_UP2 if up2 else 0)
        x2w = x2                               # what the weight gradient reads as the second tensor
        lz = lazybn.take(dy)                   # dy is d(activation) behind the BatchNorm that follows: the weight gradient runs first
        dyb = bounds.get(dy)
        b1, b2 = ctx.xb
        dy = dy.contiguous()
        B, _, H, W = dy.shape
        dev = dy.device
        L = _lib.lib()
        st = _lib.current_stream(dev)
        dx1 = dx2 = dw = db = None
        want_db = has_bias and ctx.needs_input_grad[3]
        want_w = ctx.needs_input_grad[2] or want_db
        if up2 and want_w and (b1 is None or b2 is None or (lz is None and dyb is None)):
            # an operand without a bound: the up-sampling form (fp16-split only) cannot run -- materialise the operand for this call
            x2w = torch.empty((B, C2, H, W), dtype=torch.float32, device=dev)
            with _lib.device_guard(dev):
                _lib.check(L.uaps_up_cat_fwd(x2.data_ptr(), x2.data_ptr(), x2w.data_ptr(), B, 0, C2, H // 2, W // 2, st), "uaps_up_cat_fwd")
            ucfg = cfg

        def weight_gradient(dy, dyb, lz):
            """(dw, db, dy): with a pending transform the kernel writes the true dy through (None: it has no such form here)"""
            n = C.c_size_t()
            _lib.check(L.uaps_conv_wrw_workspace_bytes(B, C1 + C2, Cout, H, W, ks, cfg, C.byref(n)), "uaps_conv_wrw_workspace_bytes")
            ws = _wrw_workspace(dev, n.value, ctx.prefs, ctx.step)
            dw = _graddest.take(ctx.keys[0], (Cout, C1 + C2, ks, ks), dev)
            db = _graddest.take(ctx.keys[1], (Cout,), dev) if want_db else None
            out = torch.empty_like(dy) if lz is not None else None
            with _timed("wrw", B, C1 + C2, Cout, H, W, ks, cfg, _h16(dyb, b1, b2), dt=lz is not None,
                        name="conv_hrwrw_up_kernel" if ucfg != cfg else None) as tm:
                hh = None
                if lz is not None:
                    hh = _lib.mk_hints((dyb, b1, b2), dyt=(lz.y, lz.coef, out, lz.slope, lz.groups))
                elif dyb is not None and b1 is not None and b2 is not None:
                    hh = _lib.mk_hints((dyb, b1, b2))
                rc = L.uaps_conv_bwd_weight_partial_cat_h(hh, dy.data_ptr(), x1.data_ptr(), C1, x2w.data_ptr(), C2, int(want_db), B, Cout,
                                                          H, W, ks, ucfg, ws.data_ptr(), ws.numel(), st)
                if lz is not None and rc == lazybn.ENOFORM:
                    tm.on = False
            if lz is not None and rc == lazybn.ENOFORM:
                return None, None, None
            _lib.check(rc, "uaps_conv_bwd_weight_partial_cat")
            _wrw_reduce(ws, dw, db, B, C1 + C2, Cout, H, W, ks, cfg, st, ctx.prefs, ctx.step)
            return dw, db, (bounds.put(out, *lz.bound) if lz is not None else dy)

        with _lib.device_guard(dev):
            done_w = False
            if lz is not None:
                if want_w and b1 is not None and b2 is not None:      # (frozen weight and bias: the stand-alone pass, no weight gradient)
                    dw, db, out = weight_gradient(dy, lz.bound, lz)
                    done_w = out is not None
                if done_w:
                    dy = out
                else:
                    dy = lazybn.materialize(dy, lz)
                dyb = bounds.get(dy)
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
                dx1 = torch.empty_like(x1)
                dx2 = torch.empty((B, C2, H, W), dtype=torch.float32, device=dev) if up2 else torch.empty_like(x2)
                with _timed("bwd_data", B, C1 + C2, Cout, H, W, ks, cfg, _h16(dyb)):
                    rc = L.uaps_conv_bwd_data_cat_h(_lib.mk_hints((dyb,)) if dyb is not None else None,
                                                    dy.data_ptr(), wb.data_ptr(), dx1.data_ptr(), C1, dx2.data_ptr(), C2, B, Cout, H, W,
                                                    ks, cfg, st)
                _lib.check(rc, "uaps_conv_bwd_data_cat")
            if want_w and not done_w:
                dw, db, _ = weight_gradient(dy, dyb, None)
            if up2 and dx2 is not None:               # the gradient of the up-sampled tensor folds to the low-resolution one (uaps_up_cat_bwd)
                dlow = torch.empty_like(x2)
                rc = L.uaps_up_cat_bwd(dx2.data_ptr(), None, dlow.data_ptr(), B, 0, C2, H // 2, W // 2, st)
                _lib.check(rc, "uaps_up_cat_bwd")
                dx2 = dlow
        return dx1, dx2, dw, db, None, None, None


def conv2d_cat(x1: torch.Tensor, x2: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
               with_stats: bool = False, stat_shift=None, up2: bool = False):
    """F.conv2d(torch.cat([x1, x2], 1), weight, bias, padding=k//2) reading the two tensors in place (stat_shift: see
    conv2d_with_stats).  up2: x2 is [B, C2, H/2, W/2] and stands for nn.Upsample(scale_factor=2, mode='bilinear',
    align_corners=True)(x2), formed in the kernels' staging (only where up2_eligible(x1, weight) and x2 carries a bound)."""
    res = _Conv2dCat.apply(x1, x2, weight, bias, with_stats, stat_shift, up2)
    return _mark(res, stat_shift is not None, weight) if with_stats else res


# ---- general strided convolutions + the stem max-pool (csrc/conv_strided.hip) --------------------------------------------------

_spacked: Dict[int, tuple] = {}


def _pack_strided(weight: torch.Tensor):
    key = id(weight)
    ent = _spacked.get(key)
    if ent is not None and ent[0]() is weight and ent[1] == weight._version and ent[2] == stamp(weight) and ent[3].device == weight.device:
        return ent[3], ent[4]
    Cout, Cin, ks, ks2 = weight.shape
    nf, nb = C.c_size_t(), C.c_size_t()
    L = _lib.lib()
    _lib.check(L.uaps_convs_pack_floats(Cout, Cin, ks, C.byref(nf), C.byref(nb)), "uaps_convs_pack_floats")
    dev = weight.device
    wf = torch.empty(nf.value, dtype=torch.float32, device=dev)
    wb = torch.empty(nb.value, dtype=torch.float32, device=dev)
    w = weight.detach().contiguous()
    with _lib.device_guard(dev):
        rc = L.uaps_convs_pack_weights(w.data_ptr(), Cout, Cin, ks, wf.data_ptr(), wb.data_ptr(), _lib.current_stream(dev))
    _lib.check(rc, "uaps_convs_pack_weights")
    _spacked[key] = (weakref.ref(weight), weight._version, stamp(weight), wf, wb)
    return wf, wb


class _ConvStrided(torch.autograd.Function):
    """F.conv2d(x, weight, None, stride, padding) for odd kernels <= 7 and stride 1 / 2 (utilities/resnet.py:120, 147, 8-14)."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding):
        _lib.require_device(x, "conv2d_strided")
        if x.dtype != torch.float32 or weight.dtype != torch.float32:
            raise TypeError("conv2d_strided: fp32 only")
        x = x.contiguous()
        B, Cin, H, W = x.shape
        Cout, Cin2, ks, ks2 = weight.shape
        if Cin2 != Cin or ks != ks2:
            raise ValueError(f"conv2d_strided: input has {Cin} channels, weight is {tuple(weight.shape)}")
        wf, wb = _pack_strided(weight)
        L = _lib.lib()
        oh, ow = C.c_int(), C.c_int()
        _lib.check(L.uaps_convs_out_size(H, W, ks, stride, padding, C.byref(oh), C.byref(ow)), "uaps_convs_out_size")
        y = torch.empty((B, Cout, oh.value, ow.value), dtype=torch.float32, device=x.device)
        with _lib.device_guard(x.device):
            rc = L.uaps_convs_fwd(x.data_ptr(), wf.data_ptr(), y.data_ptr(), B, Cin, Cout, H, W, ks, stride, padding, _lib.current_stream(x.device))
        _lib.check(rc, "uaps_convs_fwd")
        ctx.save_for_backward(x, wb)
        ctx.meta = (Cout, ks, stride, padding, id(weight))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wb = ctx.saved_tensors
        Cout, ks, stride, padding, wkey = ctx.meta
        B, Cin, H, W = x.shape
        dy = dy.contiguous()
        dev = x.device
        L = _lib.lib()
        dx = dw = None
        with _lib.device_guard(dev):
            st = _lib.current_stream(dev)
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _lib.check(L.uaps_convs_bwd_data(dy.data_ptr(), wb.data_ptr(), dx.data_ptr(), B, Cin, Cout, H, W, ks, stride, padding, st),
                           "uaps_convs_bwd_data")
            if ctx.needs_input_grad[1]:
                n = C.c_size_t()
                _lib.check(L.uaps_convs_wrw_workspace_bytes(B, Cin, Cout, H, W, ks, stride, padding, C.byref(n)), "uaps_convs_wrw_workspace_bytes")
                ws = _workspace(dev, n.value)
                dw = _graddest.take(wkey, (Cout, Cin, ks, ks), dev)
                _lib.check(L.uaps_convs_bwd_weight(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), B, Cin, Cout, H, W, ks, stride, padding,
                                                   ws.data_ptr(), ws.numel(), st), "uaps_convs_bwd_weight")
        return dx, dw, None, None


def conv2d_strided(x: torch.Tensor, weight: torch.Tensor, stride: int = 1, padding: int = 0) -> torch.Tensor:
    """Bias-free F.conv2d(x, weight, None, stride, padding), odd kernel sizes <= 7 (1, 3, 7 for the weight gradient), stride 1 or 2."""
    return _ConvStrided.apply(x, weight, int(stride), int(padding))


class _Subsample2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _lib.require_device(x, "subsample2")
        x = x.contiguous()
        B, Cc, H, W = x.shape
        y = torch.empty((B, Cc, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
        with _lib.device_guard(x.device):
            rc = _lib.lib().uaps_subsample2_fwd(x.data_ptr(), y.data_ptr(), B * Cc, H, W, _lib.current_stream(x.device))
        _lib.check(rc, "uaps_subsample2_fwd")
        ctx.shape = (B, Cc, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, Cc, H, W = ctx.shape
        dyb = bounds.get(dy)
        dy = dy.contiguous()
        dx = torch.empty((B, Cc, H, W), dtype=torch.float32, device=dy.device)
        with _lib.device_guard(dy.device):
            rc = _lib.lib().uaps_subsample2_bwd(dy.data_ptr(), dx.data_ptr(), B * Cc, H, W, _lib.current_stream(dy.device))
        _lib.check(rc, "uaps_subsample2_bwd")
        if dyb is not None:
            bounds.put(dx, *dyb)
        return dx


def subsample2(x: torch.Tensor) -> torch.Tensor:
    """x[:, :, ::2, ::2] (contiguous): what a 1x1 convolution with stride 2 samples (utilities/resnet.py:13-14)."""
    return bounds.carry(x, _Subsample2.apply(x))


class _SpaceToDepth2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _lib.require_device(x, "space_to_depth2")
        x = x.contiguous()
        B, Cc, H, W = x.shape
        xs = torch.empty((B, 4 * Cc, H // 2, W // 2), dtype=torch.float32, device=x.device)
        with _lib.device_guard(x.device):
            rc = _lib.lib().uaps_space_to_depth2(x.data_ptr(), xs.data_ptr(), B, Cc, H, W, 0, _lib.current_stream(x.device))
        _lib.check(rc, "uaps_space_to_depth2")
        ctx.shape = (B, Cc, H, W)
        return xs

    @staticmethod
    def backward(ctx, dxs):
        B, Cc, H, W = ctx.shape
        b = bounds.get(dxs)
        dxs = dxs.contiguous()
        dx = torch.empty((B, Cc, H, W), dtype=torch.float32, device=dxs.device)
        with _lib.device_guard(dxs.device):
            rc = _lib.lib().uaps_space_to_depth2(dxs.data_ptr(), dx.data_ptr(), B, Cc, H, W, 1, _lib.current_stream(dxs.device))
        _lib.check(rc, "uaps_space_to_depth2")
        if b is not None:
            bounds.put(dx, *b)
        return dx


_S2D_INDEX = {}


def _s2d_index(dev):
    """[4 phases, 9 taps] -> index into the 9 taps of the stride-2 kernel, 9 = the appended zero: along one axis the stride-2 tap
    k = 0, 1, 2 reads sampling phase (1, 0, 1) at stride-1 tap (0, 1, 1) (x[2 i + k - 1] = phase_1[i - 1], phase_0[i], phase_1[i])."""
    idx = _S2D_INDEX.get(dev)
    if idx is None:
        t = torch.full((2, 2, 3, 3), 9, dtype=torch.long)
        ph, tap = (1, 0, 1), (0, 1, 1)
        for ky in range(3):
            for kx in range(3):
                t[ph[ky], ph[kx], tap[ky], tap[kx]] = ky * 3 + kx
        idx = _S2D_INDEX[dev] = t.reshape(36).to(dev)
    return idx


def conv3x3s2_supported(x: torch.Tensor) -> bool:
    return x.is_cuda and x.shape[2] % 2 == 0 and x.shape[3] % 8 == 0


def conv3x3s2(x: torch.Tensor, weight: torch.Tensor, with_stats: bool = False):
    """F.conv2d(x, weight, None, stride=2, padding=1) for a 3x3 kernel (utilities/resnet.py:8-10, layer2.0.conv2) as a stride-1
    3x3 convolution over the four sampling phases of x stacked as channels (uaps_space_to_depth2) with the 9 taps scattered
    into a [Cout, 4 Cin, 3, 3] kernel: the same products and, per output, the same sums plus exact zeros -- on the split
    kernels of the stride-1 path, with their BatchNorm statistics epilogue.  H even, W % 8 == 0."""
    Cout, Cin = weight.shape[0], weight.shape[1]
    xs = bounds.carry(x, _SpaceToDepth2.apply(x))
    w9 = torch.cat([weight.reshape(Cout, Cin, 9), weight.new_zeros(Cout, Cin, 1)], dim=2)
    wq = w9.index_select(2, _s2d_index(weight.device)).reshape(Cout, Cin, 4, 9).permute(0, 2, 1, 3).reshape(Cout, 4 * Cin, 3, 3)
    if with_stats:
        return conv2d_with_stats(xs, wq, None)
    return conv2d(xs, wq, None)


class _MaxPool3x3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _lib.require_device(x, "maxpool3x3s2")
        x = x.contiguous()
        B, Cc, H, W = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((B, Cc, OH, OW), dtype=torch.float32, device=x.device)
        idx = torch.empty((B, Cc, OH, OW), dtype=torch.uint8, device=x.device)
        with _lib.device_guard(x.device):
            rc = _lib.lib().uaps_maxpool3x3s2_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), B * Cc, H, W, _lib.current_stream(x.device))
        _lib.check(rc, "uaps_maxpool3x3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (B, Cc, H, W)
        ctx.mark_non_differentiable(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, Cc, H, W = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((B, Cc, H, W), dtype=torch.float32, device=dy.device)
        with _lib.device_guard(dy.device):
            rc = _lib.lib().uaps_maxpool3x3s2_bwd(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), B * Cc, H, W, _lib.current_stream(dy.device))
        _lib.check(rc, "uaps_maxpool3x3s2_bwd")
        return dx


def maxpool3x3s2(x: torch.Tensor) -> torch.Tensor:
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (utilities/resnet.py:124)."""
    return bounds.carry(x, _MaxPool3x3s2.apply(x))
