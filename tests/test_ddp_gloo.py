"""world_size-2 data-parallel path on CPU (gloo): bucketed gradient averaging, identical mixing
weights on every rank, parameters in lock-step after a step.  The fused HIP loss cannot run here,
so the step's loss is supplied by the test from the oracle (tests may use the oracle as a stand-in
checker; the product default has no such fallback) and the auxiliary-decoder perturbations are
replaced by identity."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _oracle_loss(lab, y, un, w, cw1, cw2):
    from oracle import uaps_oracle as O
    from uaps_amd.losses import StepLoss
    r = O.step_loss(list(un), list(lab), y, w, cw1, cw2)
    return StepLoss(r["loss"], r["sup"], r["loss"] - r["sup"], r["pseudo"], None, None, None)


def _make_model(seed):
    from uaps_amd import unet
    torch.manual_seed(seed)
    m = unet.UNet_UAPS(3, 4, n_aux=3, feature_chns=[2, 4, 8, 16, 32], dropout=[0.0] * 5)
    ident = [lambda fs: fs] * 3
    orig = m.forward
    m.forward = lambda x, perturbations=None: orig(x, perturbations=ident)
    return m


def _batch(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return (torch.randn(2, 3, 32, 32, generator=g), torch.randint(0, 4, (2, 32, 32), generator=g), torch.randn(2, 3, 32, 32, generator=g))


def _worker(rank, world, port, overlap, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import uaps_amd
    model = _make_model(seed=rank)                 # different init per rank: broadcast must fix it
    uaps_amd.dist.broadcast_model(model)
    tr = uaps_amd.UAPSTrainer(model, loss_fn=_oracle_loss, overlap_comm=overlap, seed=5)
    grads = {}
    xl, yl, xu = _batch(rank)
    res = tr.train_step(xl, yl, xu)
    torch.save({"w": res["w"], "grads": {n: p.grad.clone() for n, p in model.named_parameters()},
                "params": {k: v.clone() for k, v in model.state_dict().items()}, "buckets": tr.buckets.names},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_two_rank_step_matches_single_process_average(tmp_path, overlap):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, overlap, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"rank{i}.pt"), weights_only=False) for i in range(world)]
    assert np.array_equal(r[0]["w"], r[1]["w"])                         # same Dirichlet draw on every rank
    assert r[0]["buckets"][-1] == "encoder" and r[0]["buckets"][0] == "aux_decoder3"
    for n in r[0]["grads"]:
        assert torch.equal(r[0]["grads"][n], r[1]["grads"][n]), n       # all-reduced gradients identical
    for k in r[0]["params"]:
        if "running_" in k or "num_batches" in k:
            continue                                                     # BN buffers stay rank-local, like DataParallel replicas
        assert torch.equal(r[0]["params"][k], r[1]["params"][k]), k
    # single-process reference: average of the two ranks' gradients computed one after the other
    sys.path.insert(0, ROOT)
    import uaps_amd
    torch.set_num_threads(1)
    ref = None
    for rank in range(world):
        m = _make_model(seed=0)
        tr = uaps_amd.UAPSTrainer(m, loss_fn=_oracle_loss, seed=5)
        captured = {}
        hooks = [p.register_hook(lambda g, n=n: captured.__setitem__(n, g.clone())) for n, p in m.named_parameters()]
        tr.train_step(*_batch(rank))
        ref = captured if ref is None else {n: (ref[n] + captured[n]) / 2 for n in ref}
    for n, gr in ref.items():
        torch.testing.assert_close(r[0]["grads"][n], gr, rtol=1e-5, atol=1e-7, msg=n)
