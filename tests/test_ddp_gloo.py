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


def _oracle_loss(lab, y, un, w, cw1, cw2, exchange=None):
    """Stand-in for the HIP loss block on the CPU.  With `exchange` it follows the product's gathered-batch protocol
    (losses._PairLoss): raw sums of this shard -> summed over the ranks -> loss from the global sums and pixel count; the
    gradient flows through this rank's own contribution to the sums."""
    from oracle import uaps_oracle as O
    from uaps_amd.losses import StepLoss
    if exchange is None:
        r = O.step_loss(list(un), list(lab), y, w, cw1, cw2)
        return StepLoss(r["loss"], r["sup"], r["loss"] - r["sup"], r["pseudo"], None, None, None)
    t = O.loss_sums(list(un), list(lab), y, w)
    keys = [k for k in t if k != "pseudo"]
    flat = torch.cat([t[k].reshape(-1).double() for k in keys])
    summed = flat.detach().clone()
    world = exchange(summed)
    flat = flat + (summed - flat.detach())                 # value = global sums, derivative = d(local sums)
    tot, off = {}, 0
    for k in keys:
        n = t[k].numel()
        tot[k] = flat[off:off + n].view_as(t[k]).to(t[k].dtype)
        off += n
    n_pix = y.numel() * world
    r = O.loss_from_sums(tot, n_pix, cw1, cw2)
    return StepLoss(r["loss"], r["sup"], r["loss"] - r["sup"], t["pseudo"], None, None, None)


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


def _worker(rank, world, port, overlap, out_dir, gathered=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import uaps_amd
    model = _make_model(seed=rank)                 # different init per rank: broadcast must fix it
    uaps_amd.dist.broadcast_model(model)
    tr = uaps_amd.UAPSTrainer(model, loss_fn=_oracle_loss, overlap_comm=overlap, seed=5, gathered_loss=gathered)
    grads = {}
    xl, yl, xu = _batch(rank)
    res = tr.train_step(xl, yl, xu)
    torch.save({"w": res["w"], "loss": float(res["loss"]), "grads": {n: p.grad.clone() for n, p in model.named_parameters()},
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


def test_four_ranks_bucket_order_and_average(tmp_path):
    """world_size 4: the buckets follow the order in which the product's backward finishes them (auxiliary decoders last
    created first, encoder last), every rank ends with identical averaged gradients and parameters."""
    world, port = 4, _free_port()
    mp.spawn(_worker, args=(world, port, True, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"rank{i}.pt"), weights_only=False) for i in range(world)]
    assert r[0]["buckets"] == ["aux_decoder3", "aux_decoder2", "aux_decoder1", "main_decoder", "encoder"]
    for i in range(1, world):
        assert np.array_equal(r[0]["w"], r[i]["w"])
        for n in r[0]["grads"]:
            assert torch.equal(r[0]["grads"][n], r[i]["grads"][n]), n
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
        ref = captured if ref is None else {n: ref[n] + captured[n] for n in ref}
    for n, gr in ref.items():
        torch.testing.assert_close(r[0]["grads"][n], gr / world, rtol=1e-5, atol=1e-7, msg=n)


def test_gathered_loss_two_ranks_equals_the_gathered_batch(tmp_path):
    """gathered_loss=True (SURVEY 8e's optional exactness step): the loss statistics run over the batch of both ranks, as
    on the reference's gathered logits (UAPS_model.py:13 + UAPS_train.py:194-277): the loss value on every rank and the
    SUMMED gradients must equal a single process that forwards each shard (per-replica BatchNorm statistics, as
    DataParallel replicas have) and computes one loss on the concatenated logits."""
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, True, str(tmp_path), True), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"rank{i}.pt"), weights_only=False) for i in range(world)]
    assert abs(r[0]["loss"] - r[1]["loss"]) < 1e-7
    for n in r[0]["grads"]:
        assert torch.equal(r[0]["grads"][n], r[1]["grads"][n]), n
    sys.path.insert(0, ROOT)
    import uaps_amd
    from oracle import uaps_oracle as O
    torch.set_num_threads(1)
    m = _make_model(seed=0)
    tr = uaps_amd.UAPSTrainer(m, loss_fn=_oracle_loss, seed=5)
    m.train()
    labs, uns, ys = [], [], []
    for rank in range(world):
        xl, yl, xu = _batch(rank)
        labs.append(m(xl)); uns.append(m(xu)); ys.append(yl)
    D = len(labs[0])
    w = tr.mix_rng.dirichlet(np.ones(D), size=1)[0]
    cw1, cw2 = tr.consistency_weights()
    lab_all = [torch.cat([l[k] for l in labs]) for k in range(D)]
    un_all = [torch.cat([u[k] for u in uns]) for k in range(D)]
    ref = O.step_loss(un_all, lab_all, torch.cat(ys), w, cw1, cw2)
    ref["loss"].backward()
    assert np.array_equal(w, r[0]["w"])
    np.testing.assert_allclose(r[0]["loss"], float(ref["loss"]), rtol=1e-6)
    for n, p in m.named_parameters():
        torch.testing.assert_close(r[0]["grads"][n], p.grad, rtol=2e-4, atol=1e-7, msg=n)
