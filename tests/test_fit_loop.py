"""The epoch loop around the step (UAPSTrainer.fit) against the conventions of the reference's Network.run()
(UAPS_train.py:127-159, 279-280, 316-321, 367-402, 427-450; the cycle() form of DAGM-Dataset-codes/UAPS_train.py:143).
The device work (train_step / validate) is scripted here, the loop logic is what is under test; the real thing runs in
tests/test_gpu_trainer.py."""
import os

import numpy as np
import torch

import uaps_amd
from uaps_amd.ramps import get_current_consistency_weight


class _Scripted(uaps_amd.UAPSTrainer):
    def __init__(self, val_dice, **kw):
        net = uaps_amd.UNet_UAPS(3, 4, feature_chns=[2, 2, 2, 2, 2])
        super().__init__(net, loss_fn=lambda *a: None, **kw)
        self.val_dice, self.seen, self.val_calls, self.cw_log = list(val_dice), [], 0, []

    def train_step(self, x_l, y_l, x_u, w=None):
        self.seen.append((float(x_l.flatten()[0]), float(x_u.flatten()[0])))
        self.cw_log.append(self.consistency_weights()[0])
        self.iter_num += 1
        v = torch.tensor(float(self.iter_num))
        return {"loss": v, "sup": v * 2, "unsup": v * 3}

    def epoch_metrics(self, reset=True, pooled=False):
        return {"miou": 0.5, "mdice": 0.25, "acc": 1.0}

    def validate(self, batches, pooled=False):
        d = self.val_dice[self.val_calls]
        self.val_calls += 1
        return {"miou": d / 2, "mdice": d, "acc": 1.0, "ce": 0.1, "dice_loss": 1 - d, "loss": 0.5 * (1 - d + 0.1)}


def _loaders(n_l, n_u):
    lab = [(torch.full((1, 3, 4, 4), float(i)), torch.zeros(1, 4, 4, dtype=torch.long)) for i in range(n_l)]
    unl = [(torch.full((1, 3, 4, 4), 100.0 + i), torch.zeros(1, 4, 4, dtype=torch.long)) for i in range(n_u)]
    return lab, unl


def test_fit_follows_the_reference_loop(tmp_path):
    dice = [0.30, 0.30, 0.45, 0.40]
    tr = _Scripted(dice)
    lab, unl = _loaders(2, 3)
    path = os.path.join(tmp_path, "Checkpoints", "UAPS.pth")
    hist = tr.fit(lab, unl, [None], epochs=5, iter_per_epoch=6, checkpoint_path=path)
    # range(1, epochs) epochs of range(1, iter_per_epoch) steps
    assert [h["epoch"] for h in hist] == [1, 2, 3, 4] and tr.iter_num == 4 * 5
    # a fresh zip(cycle(labelled), cycle(unlabelled)) every epoch: the shorter labelled loader is over-sampled, no StopIteration
    assert tr.seen[:5] == [(0.0, 100.0), (1.0, 101.0), (0.0, 102.0), (1.0, 100.0), (0.0, 101.0)]
    assert tr.seen[5:10] == tr.seen[:5]
    # the logged means divide the running sums by iter_per_epoch, not by the number of steps (:316)
    assert np.isclose(hist[0]["loss"], sum(range(1, 6)) / 6) and np.isclose(hist[1]["unsup"], 3 * sum(range(6, 11)) / 6)
    assert np.isclose(hist[0]["train_mdice"], 0.25 * 5 / 6)
    # strictly greater validation mDice saves (:427), anything else counts patience
    assert [h["saved"] for h in hist] == [True, False, True, False]
    assert [h["patience"] for h in hist] == [0, 1, 0, 1] and hist[-1]["best_dice"] == 0.45
    ck = torch.load(path, weights_only=False)
    assert ck["epoch"] == 3 and ck["best_dice_1"] == 0.45 and ck["iter_num"] == 15
    assert all(k.startswith("module.") for k in ck["state_dict"])
    # scheduler.step(val mDice) once per epoch (:402), mode "max"
    assert tr.scheduler.last_epoch == 4 and np.isclose(tr.scheduler.best, 0.45)


def test_ramp_runs_on_iter_num_across_epochs():
    tr = _Scripted([0.1] * 3, consistency_rampup=4, ramp_divisor=3)
    lab, unl = _loaders(1, 1)
    tr.fit(lab, unl, [None], epochs=4, iter_per_epoch=5)
    want = [get_current_consistency_weight(0.1, i, 4, 3) for i in range(12)]
    assert np.allclose(tr.cw_log, want) and tr.cw_log[0] < tr.cw_log[5] < tr.cw_log[11]


def test_fit_resumes_mid_run(tmp_path):
    dice = [0.2, 0.5, 0.4, 0.6]
    lab, unl = _loaders(2, 2)
    path = os.path.join(tmp_path, "ck.pth")
    full = _Scripted(dice)
    h_full = full.fit(lab, unl, [None], epochs=5, iter_per_epoch=4, checkpoint_path=path + ".full")
    first = _Scripted(dice)
    first.fit(lab, unl, [None], epochs=3, iter_per_epoch=4, checkpoint_path=path)
    ck = torch.load(path, weights_only=False)
    assert ck["epoch"] == 2 and ck["iter_num"] == 6
    second = _Scripted(dice[2:])
    got = second.load_checkpoint(path)
    assert second.iter_num == 6 and second.scheduler.last_epoch == 2
    h2 = second.fit(lab, unl, [None], epochs=5, iter_per_epoch=4, start_epoch=got["epoch"] + 1, best_dice=got["best_dice_1"],
                    checkpoint_path=path)
    assert [h["epoch"] for h in h2] == [3, 4] and second.iter_num == full.iter_num
    assert [h["saved"] for h in h2] == [h["saved"] for h in h_full[2:]] == [False, True]
    assert second.cw_log == full.cw_log[6:]
    # the Dirichlet stream of the mixing weights continues where the checkpoint left it
    assert np.array_equal(second.mix_rng.get_state()[1], first.mix_rng.get_state()[1])


def test_gradient_destination_is_handed_out_once_per_backward():
    """A parameter used by two autograd nodes of one backward (two forwards of one model: a ragged last batch) must not get
    the same bucket slice twice: the second node would overwrite the first node's gradient and autograd would sum two aliases."""
    from uaps_amd import _graddest
    p = torch.nn.Parameter(torch.zeros(3, 2))
    flat = torch.zeros(8)
    _graddest.register(p, flat, 0)
    try:
        a = _graddest.take(id(p), (3, 2), flat.device)
        b = _graddest.take(id(p), (3, 2), flat.device)
        assert a.data_ptr() == flat.data_ptr() and b.data_ptr() != flat.data_ptr()
        a.fill_(1.0); b.fill_(10.0)
        assert float((a + b).sum()) == 66.0 and float(flat[:6].sum()) == 6.0
        _graddest.new_backward()
        assert _graddest.take(id(p), (3, 2), flat.device).data_ptr() == flat.data_ptr()
        assert _graddest.take(id(p), (2, 3), flat.device).data_ptr() != flat.data_ptr()      # shape mismatch: never the slice
    finally:
        _graddest.unregister(p)
        _graddest.new_backward()
