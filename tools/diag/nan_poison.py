#!/usr/bin/env python3
"""Does any kernel of the step read memory nobody wrote?  torch.empty is made to fill new tensors with NaN (floats) / the largest
integer (torch.utils.deterministic.fill_uninitialized_memory), a few steps are run, and the losses / parameters are compared
with the unpoisoned run.   python tools/diag/nan_poison.py [streams]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import copy
import numpy as np
import torch
import uaps_amd
from uaps_amd import unet, perturb

DEV = torch.device("cuda:0")


def run(poison, streams, steps=6, kw=None):
    unet._DECODER_STREAMS = streams
    torch.manual_seed(0)
    m = unet.UNet_UAPS(3, 4, n_aux=3, feature_chns=[8, 16, 32, 64, 128]).to(DEV)
    tr = uaps_amd.UAPSTrainer(m, seed=5, **(kw or {}))
    perturb.manual_seed(5, 0); np.random.seed(5)
    g = torch.Generator().manual_seed(3)
    data = [(torch.randn(2, 3, 64, 64, generator=g).to(DEV), torch.randint(0, 4, (2, 64, 64), generator=g).to(DEV),
             torch.randn(2, 3, 64, 64, generator=g).to(DEV)) for _ in range(2)]
    torch.use_deterministic_algorithms(poison, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = poison
    losses = []
    try:
        for i in range(steps):
            losses.append(float(tr.train_step(*data[i % 2])["loss"]))
    finally:
        torch.use_deterministic_algorithms(False)
    return losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


if __name__ == "__main__":
    for streams in (False, True):
        for kw in ({}, {"step_state": True}, {"use_graph": True}):
            a, pa = run(False, streams, kw=kw)
            b, pb = run(True, streams, kw=kw)
            bad = [k for k in pa if not torch.equal(pa[k], pb[k]) and not (torch.isnan(pa[k]).any() and torch.isnan(pb[k]).any())]
            print(f"streams={streams} {kw}: clean {a}  poisoned {b}  differing tensors {len(bad)} {bad[:4]}", flush=True)
