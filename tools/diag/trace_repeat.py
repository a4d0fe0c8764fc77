#!/usr/bin/env python3
"""Which launch deviates first?  Every autograd Function of the package is wrapped so that a bit hash of each tensor it returns
(forward and backward) is appended to a per-run trace; `ref` saves the trace of one solo run, `check N` repeats the run N times
(start two at once to share the card) and reports the first trace entry that differs from the solo reference.
  python tools/diag/trace_repeat.py ref /tmp/t.pt ; python tools/diag/trace_repeat.py check 8 /tmp/t.pt & (x2)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import uaps_amd
from uaps_amd import unet, perturb, conv, fused, losses
import test_gpu_two_ranks as T

TRACE, NAMES = [], []


def _hash(t):
    if t.dtype == torch.float32:
        return t.contiguous().view(torch.int32).sum(dtype=torch.int64)
    return t.contiguous().to(torch.int64).sum()


def _record(name, res):
    outs = res if isinstance(res, (tuple, list)) else (res,)
    for i, o in enumerate(outs):
        if torch.is_tensor(o) and o.is_cuda and o.numel() > 0:
            TRACE.append(_hash(o.detach())); NAMES.append(f"{name}[{i}] {tuple(o.shape)}")


def _wrap_all():
    for mod in (conv, fused, perturb, losses):
        for k, v in list(vars(mod).items()):
            if isinstance(v, type) and issubclass(v, torch.autograd.Function) and v is not torch.autograd.Function:
                for which in ("forward", "backward"):
                    f = getattr(v, which)
                    def make(f, label):
                        def g(*a, **kw):
                            r = f(*a, **kw)
                            if label.endswith("forward") and getattr(a[0], "to_save", None):
                                _record(label + ".saved", tuple(a[0].to_save))      # what the forward kept (xf, stats ...) first
                            _record(label, r)
                            return r
                        return staticmethod(g)
                    setattr(v, which, make(f, f"{mod.__name__.split('.')[-1]}.{k}.{which}"))


def run(steps):
    unet._DECODER_STREAMS = os.environ.get("UAPS_TEST_STREAMS", "0") != "0"
    model = T._make_model(seed=0)
    tr = uaps_amd.UAPSTrainer(model, seed=T.SEED, step_state=True)
    perturb.manual_seed(T.SEED, 0); np.random.seed(T.SEED)
    TRACE.clear(); NAMES.clear()
    marks = []
    for s in range(steps):
        tr.train_step(*T._batch(0, s % 3))
        with torch.no_grad():
            for n, p in model.named_parameters():
                TRACE.append(_hash(p.detach())); NAMES.append(f"param {n}")
        marks.append(len(TRACE))
    torch.cuda.synchronize()
    return torch.stack(TRACE).cpu(), list(NAMES), marks


if __name__ == "__main__":
    _wrap_all()
    steps = int(os.environ.get("TR_STEPS", "20"))
    if sys.argv[1] == "ref":
        h, names, marks = run(steps)
        torch.save({"h": h, "names": names, "marks": marks}, sys.argv[2])
        print(f"reference trace: {len(names)} entries over {steps} steps")
    else:
        n, ref = int(sys.argv[2]), torch.load(sys.argv[3])
        bad = 0
        for i in range(n):
            h, names, marks = run(steps)
            assert names == ref["names"]
            if not torch.equal(h, ref["h"]):
                bad += 1
                d = (h != ref["h"]).nonzero().flatten()
                first = int(d[0])
                step = next(s for s, m in enumerate(marks) if first < m)
                base = marks[step - 1] if step else 0
                # entries of that step that differ, in order (the first is the culprit's output, the rest its consequences)
                here = [int(j) for j in d if j < marks[step]][:6]
                print(f"pid {os.getpid()} run {i}: first difference at entry {first} = step {step} + {first - base}: {names[first]};"
                      f" next: {[names[j] for j in here[1:]]}; {len(d)} entries differ in all", flush=True)
        print(f"pid {os.getpid()}: {n} runs, {bad} deviated", flush=True)
