#!/usr/bin/env python3
"""Single-process repeatability: the same 20 training steps (tests/test_gpu_two_ranks.py model and batches, decoder streams)
run N times in one process -- eager state mode and captured -- and the bit hashes of all parameters after every step compared
with the first run's.   python tools/diag/sp_repeat.py [runs] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import uaps_amd
from uaps_amd import unet, perturb
import test_gpu_two_ranks as T


def run(kw, steps):
    unet._DECODER_STREAMS = os.environ.get("UAPS_TEST_STREAMS", "1") != "0"
    model = T._make_model(seed=0)
    tr = uaps_amd.UAPSTrainer(model, seed=T.SEED, **kw)
    perturb.manual_seed(T.SEED, 0); np.random.seed(T.SEED)
    hs = []
    for s in range(steps):
        tr.train_step(*T._batch(0, s % 3))
        with torch.no_grad():
            hs.append(torch.stack([p.detach().view(torch.int32).sum(dtype=torch.int64) for p in model.parameters()]))
    torch.cuda.synchronize()
    return torch.stack(hs).cpu()


if __name__ == "__main__":
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    names = [n for n, _ in T._make_model(seed=0).named_parameters()]
    eager_ref = None
    for name, kw in (("eager state", {"step_state": True}), ("graph", {"use_graph": True})):
        ref, bad, vs_eager = None, [], []
        for i in range(runs):
            h = run(kw, steps)
            if ref is None:
                ref = h
            elif not torch.equal(ref, h):
                st = int((ref != h).any(dim=1).nonzero()[0])
                bad.append((i, st))
            if eager_ref is not None and not torch.equal(eager_ref, h):
                st = int((eager_ref != h).any(dim=1).nonzero()[0])
                cols = (eager_ref[st] != h[st]).nonzero().flatten().tolist()
                vs_eager.append((i, st, len(cols), [names[c] for c in cols[:3]]))
        if name == "eager state":
            eager_ref = ref
        print(f"{name}: {runs} runs of {steps} steps, differing from the first: {bad or 'none'}; differing from eager: {vs_eager or 'none'}", flush=True)
