"""Two trainers driven from two Python threads.

The host-side state of a step -- the queue of deferred weight-gradient reductions, the lazy BatchNorm backward's count of pending
records, the forward's scratch -- is owned by the step (uaps_amd/stepctx.py), not by module globals: two threads that each step a
model of their own, with deferred reductions and the two-halves BatchNorm backward ON, must each end exactly where a
single-threaded run of the same steps ends, bit for bit.  What a thread has to bring itself: a HIP stream of its own (workspaces
are cached per stream) and, because the perturbation draws come from one process-wide Philox stream as the reference's come from
one global RNG, a private stream of draws (perturb.local_rng)."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(seed: int, steps: int):
    """Model and batches of one trainer, built on the CALLING thread: parameter initialisation draws from torch's process-wide CPU
    generator, which two threads must not share."""
    import uaps_amd
    torch.manual_seed(seed)
    model = uaps_amd.net_factory("unet_uaps", 3, 4, n_aux=2)          # FeatureNoise + Dropout decoders (FeatureDropout's threshold is numpy's global RNG)
    g = torch.Generator().manual_seed(1000 + seed)
    batches = [(torch.randn(2, 3, 32, 256, generator=g).to(DEV), torch.randint(0, 4, (2, 32, 256), generator=g).to(DEV),
                torch.randn(2, 3, 32, 256, generator=g).to(DEV)) for _ in range(steps)]
    torch.cuda.synchronize()
    return model, batches


def _run(seed: int, model, batches, barrier=None, out=None, key=None):
    import uaps_amd
    from uaps_amd import lazybn, perturb, stepctx
    stream = torch.cuda.Stream(device=DEV)
    res = None
    try:
        with torch.cuda.stream(stream), perturb.local_rng(seed):
            tr = uaps_amd.UAPSTrainer(model, seed=seed, track_metrics=False)
            if barrier is not None:
                barrier.wait()
            n0 = lazybn.prepared_total()
            losses = []
            for xl, y, xu in batches:
                losses.append(tr.train_step(xl, y, xu)["loss"])
                assert stepctx.current() is None and lazybn.current() is None
            stream.synchronize()
            tr.check_errors()
            res = {"loss": [float(v) for v in losses], "lazy": lazybn.prepared_total() - n0,
                   "params": {n: p.detach().cpu().numpy() for n, p in model.named_parameters()}}
    except BaseException as e:                      # a thread's exception must reach the test
        res = e
    if out is not None:
        out[key] = res
    return res


def test_two_trainers_in_two_threads_equal_their_single_threaded_runs():
    from uaps_amd import conv
    if conv.get_mode() != "h16":
        pytest.skip("the two-halves BatchNorm backward and the row kernels it rides on exist in the default arithmetic")
    assert conv._DEFER
    steps = 4
    solo = {s: _run(s, *_setup(s, steps)) for s in (3, 4)}
    for s in solo.values():
        assert not isinstance(s, BaseException), s
        assert s["lazy"] > 0                         # 256-wide maps: the lazy BatchNorm backward did run
    out, bar = {}, threading.Barrier(2)
    fresh = {s: _setup(s, steps) for s in (3, 4)}                  # the same initial parameters and batches again
    ts = [threading.Thread(target=_run, args=(s, *fresh[s], bar, out, s)) for s in (3, 4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    for s in (3, 4):
        assert s in out and not isinstance(out[s], BaseException), out.get(s)
        assert out[s]["loss"] == solo[s]["loss"]
        for n, a in solo[s]["params"].items():
            np.testing.assert_array_equal(out[s]["params"][n], a, err_msg=f"trainer {s}: {n}")
