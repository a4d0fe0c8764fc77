"""torch.optim.Adam with the step executed by one hand-written multi-tensor HIP kernel (csrc/adam.hip).

Drop-in for the reference's `torch.optim.Adam(model.parameters(), lr=args.base_lr)` (UAPS_train.py:112): same
constructor, same `state_dict()` layout (`step`, `exp_avg`, `exp_avg_sq` per parameter; the checkpoint dict of
UAPS_train.py:443-448 stores it), same arithmetic order as PyTorch's single-tensor Adam, so a checkpoint written by
either loads into the other.  amsgrad / maximize / capturable / differentiable are refused (the reference uses none)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, **kw):
        if amsgrad or kw.get("maximize") or kw.get("capturable") or kw.get("differentiable"):
            raise ValueError("uaps_amd.optim.Adam: amsgrad / maximize / capturable / differentiable are not supported")
        kw.pop("fused", None); kw.pop("foreach", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, foreach=False, fused=False, **kw)
        # True: the kernel reads lr / (1 - beta1^t) and 1 / sqrt(1 - beta2^t) from the step state (uaps_set_step_state) instead
        # of taking them by value, so that a captured step can be replayed (trainer.StepGraph keeps the state current)
        self.from_step_state = False
        self._early_done = set()                 # ids of the parameters step_early has taken since the last step()

    def step_scalars(self, t: int):
        """(lr / bias_correction1, 1 / sqrt(bias_correction2)) of step t for the first parameter group, as uaps_adam_step computes them."""
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        return float(g["lr"]) / (1.0 - b1 ** t), 1.0 / (1.0 - b2 ** t) ** 0.5

    def _launch(self, group, ps) -> None:
        """One uaps_adam_step over the parameters `ps` (with gradients) of `group`."""
        L = _lib.lib()
        dev = ps[0].device
        _lib.require_device(ps[0], "uaps_amd.optim.Adam")
        grads, ms, vs, steps = [], [], [], []
        for p in ps:
            if p.dtype != torch.float32 or p.device != dev or not p.is_contiguous():
                raise TypeError("uaps_amd.optim.Adam: contiguous float32 parameters on one device expected")
            st = self.state[p]
            if len(st) == 0:                                  # same lazy state as torch.optim.Adam (fused=False)
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            grads.append(g); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"]); steps.append(st["step"])
        torch._foreach_add_(steps, 1)                         # host-resident step counters, one call
        step = int(steps[0])
        if int(steps[-1]) != step:
            raise RuntimeError("uaps_amd.optim.Adam: parameters of one group must share the step count")
        if self.from_step_state:                              # lr / bias corrections come from the device step state
            step = 0
        n = len(ps)
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        sizes = (C.c_long * n)(*[p.numel() for p in ps])
        b1, b2 = group["betas"]
        lr = float(group["lr"])
        with _lib.device_guard(dev):
            rc = L.uaps_adam_step(arr(ps), arr(grads), arr(ms), arr(vs), sizes, n, lr, float(b1), float(b2), float(group["eps"]),
                                  float(group["weight_decay"]), step, _lib.current_stream(dev))
        _lib.check(rc, "uaps_adam_step")

    @torch.no_grad()
    def step_early(self, params) -> int:
        """The step of the parameters in `params` only (those that have a gradient), on the current stream; the following step()
        leaves them out.  For a caller that knows these gradients to be final while the rest of the backward still runs
        (UAPSTrainer: the decoders' parameters, beside the encoder's backward).  Returns the number of parameters stepped."""
        want = {id(p) for p in params}
        n = 0
        for group in self.param_groups:
            ps = [p for p in group["params"] if id(p) in want and p.grad is not None and id(p) not in self._early_done]
            if ps:
                self._launch(group, ps)
                self._early_done.update(id(p) for p in ps)
                n += len(ps)
        return n

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        done, self._early_done = self._early_done, set()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None and id(p) not in done]
            if ps:
                self._launch(group, ps)
            if done and ps:
                # the early half and this half are ONE optimizer step: a parameter of each must now carry the same count (a step
                # abandoned between its halves would leave the early parameters one ahead for good: ADVICE r5)
                early = next((p for p in group["params"] if id(p) in done and p in self.state), None)
                if early is not None and int(self.state[early]["step"]) != int(self.state[ps[0]]["step"]):
                    raise RuntimeError(f"uaps_amd.optim.Adam: step counts {int(self.state[early]['step'])} (early half) and "
                                       f"{int(self.state[ps[0]]['step'])} (final half) differ: a previous step was abandoned between its halves")
        return loss
