"""-m gpu: the step-state form of the training step (uaps_amd/graph.py) and its hipGraph capture.

1. state mode == by-value mode once the random draws are taken out (identity perturbations, dropout p = 0): the Dirichlet
   weights, consistency weights and Adam scalars read from the device step state give the same step as the same numbers
   passed as kernel arguments.
2. replaying the captured step == running the same state-mode step eagerly, bit for bit, perturbations included (both
   draw from Philox key + per-step key increment with the same counters), over warm-up, capture and replays.
3. the device-drawn FeatureDropout threshold lies in [0.7, 0.9) x max attention and changes from step to step.
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(seed, widths=(8, 16, 16, 32, 32)):
    import uaps_amd
    torch.manual_seed(seed)
    return uaps_amd.UNet_UAPS(3, 4, feature_chns=list(widths))


def _batches(n, B, H, W, seed=5):
    import uaps_amd
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        xl = torch.tensor(rng.standard_normal((B, 3, H, W)).astype(np.float32)).to(DEV)
        xu = torch.tensor(rng.standard_normal((B, 3, H, W)).astype(np.float32)).to(DEV)
        y = torch.tensor(uaps_amd.data.synthetic_masks(rng, B, 4, H, W)).to(DEV)
        out.append((xl, y, xu))
    return out


def test_state_mode_equals_by_value_mode_without_random_draws():
    import uaps_amd
    m0 = _model(3)
    for m in m0.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    m1 = copy.deepcopy(m0)
    ident = [lambda fs: fs] * 3
    trainers = []
    for m, kw in ((m0, {}), (m1, {"step_state": True})):
        m.to(DEV)
        orig = m.forward_pair
        m.forward_pair = (lambda a, b, perturbations=None, _o=orig: _o(a, b, perturbations=ident))
        trainers.append(uaps_amd.UAPSTrainer(m, base_lr=1e-3, seed=11, **kw))
    assert trainers[1].step_graph is not None and trainers[0].step_graph is None
    for xl, y, xu in _batches(3, 2, 64, 64):
        ra = trainers[0].train_step(xl, y, xu)
        rb = trainers[1].train_step(xl, y, xu)
        np.testing.assert_allclose(float(rb["loss"]), float(ra["loss"]), rtol=1e-6)
        np.testing.assert_allclose(rb["w"], ra["w"])
    for (n, pa), pb in zip(m0.named_parameters(), m1.parameters()):
        # identical gradients; Adam's lr/bias-correction scalars are rounded to fp32 on the host instead of in the launcher
        np.testing.assert_allclose(pb.detach().cpu().numpy(), pa.detach().cpu().numpy(), rtol=0, atol=2e-7, err_msg=n)
    assert trainers[0].epoch_metrics() == trainers[1].epoch_metrics()


@pytest.mark.parametrize("streams", [False, True])
def test_graph_replay_is_bit_identical_to_the_eager_state_step(streams, monkeypatch):
    import uaps_amd
    m0 = _model(4)
    m1 = copy.deepcopy(m0)
    m0.to(DEV), m1.to(DEV)
    from uaps_amd import unet
    monkeypatch.setattr(unet, "_DECODER_STREAMS", streams)
    eager = uaps_amd.UAPSTrainer(m0, base_lr=1e-3, seed=7, step_state=True)
    graph = uaps_amd.UAPSTrainer(m1, base_lr=1e-3, seed=7, use_graph=True)
    losses_e, losses_g = [], []
    for i, (xl, y, xu) in enumerate(_batches(6, 2, 64, 64)):
        uaps_amd.perturb.manual_seed(7, 0)
        np.random.seed(7)
        losses_e.append(eager.train_step(xl, y, xu)["loss"].clone())
        uaps_amd.perturb.manual_seed(7, 0)
        np.random.seed(7)
        losses_g.append(graph.train_step(xl, y, xu)["loss"].clone())
        assert (graph.step_graph.graph is not None) == (i >= 2)          # two eager warm-up steps, then capture + replays
    torch.cuda.synchronize()
    assert [float(a) for a in losses_e] == [float(b) for b in losses_g]
    assert len({float(a) for a in losses_e}) == 6
    for (n, pa), pb in zip(m0.named_parameters(), m1.parameters()):
        assert torch.equal(pa, pb), n
    for (n, ba), bb in zip(m0.named_buffers(), m1.buffers()):
        assert torch.equal(ba, bb), n
    sa, sb = eager.optimizer.state_dict()["state"], graph.optimizer.state_dict()["state"]
    assert all(float(sa[k]["step"]) == float(sb[k]["step"]) == 6.0 for k in sa)
    assert all(torch.equal(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"]) for k in sa)
    assert eager.epoch_metrics() == graph.epoch_metrics()
    assert eager.iter_num == graph.iter_num == 6
    # a different batch shape falls back to the eager state-mode step, and the graph keeps working afterwards
    (xl, y, xu), = _batches(1, 1, 64, 64, seed=9)
    graph.train_step(xl, y, xu)
    (xl, y, xu), = _batches(1, 2, 64, 64, seed=10)
    r = graph.train_step(xl, y, xu)
    assert np.isfinite(float(r["loss"]))


def test_device_drawn_feature_dropout_threshold():
    """FeatureDropout (UAPS_unet.py:156-169): threshold = max attention x U(0.7, 0.9), pixels whose channel mean reaches it are
    zeroed.  With the step state active the U is a Philox draw on the device: per image, the attention ratios of the kept and
    the dropped pixels must be separable by one u in [0.7, 0.9), and that u changes with the step key."""
    from uaps_amd import perturb
    from uaps_amd.graph import StepState
    torch.manual_seed(0)
    f = torch.rand(4, 16, 32, 32, device=DEV) + 0.1
    att = f.mean(1)
    r = att / att.flatten(1).max(1).values.view(-1, 1, 1)
    st = StepState(torch.device(DEV))
    brackets = []
    prev = perturb.DEVICE_THRESHOLDS
    perturb.DEVICE_THRESHOLDS = True
    try:
        for _ in range(2):
            st.fill([1.0], 0.0, 0.0, 1e-3, 1.0)
            st.upload()
            st.activate()
            perturb._RngState.offset = 0
            _, out = perturb.perturbed_fan_out(f, ["feature_dropout"], 1)
            StepState.deactivate()
            kept = (out != 0).any(1)
            assert bool(kept.any()) and bool((~kept).any())
            top_kept = torch.where(kept, r, torch.full_like(r, -1.0)).flatten(1).max(1).values
            low_drop = torch.where(~kept, r, torch.full_like(r, 9.0)).flatten(1).min(1).values
            assert bool((top_kept < low_drop).all())                    # a threshold separates them
            assert bool((low_drop >= 0.7).all()) and bool((top_kept < 0.9).all())
            brackets.append((top_kept.cpu(), low_drop.cpu()))
    finally:
        perturb.DEVICE_THRESHOLDS = prev
        StepState.deactivate()
    assert not (torch.equal(brackets[0][0], brackets[1][0]) and torch.equal(brackets[0][1], brackets[1][1]))


def test_validation_between_replays_leaves_the_captured_step_intact():
    """An eval-mode forward between replays (UAPSTrainer.validate: other packed-weight buffers, no statistics groups, no bounds)
    must neither disturb the graph nor read stale weights: the eager state-mode trainer doing the same sequence stays
    bit-identical, and both validations agree."""
    import uaps_amd
    m0 = _model(8)
    m1 = copy.deepcopy(m0)
    m0.to(DEV), m1.to(DEV)
    eager = uaps_amd.UAPSTrainer(m0, base_lr=1e-3, seed=5, step_state=True)
    graph = uaps_amd.UAPSTrainer(m1, base_lr=1e-3, seed=5, use_graph=True)
    batches = _batches(7, 2, 64, 64, seed=21)
    val = [(b[0], b[1]) for b in _batches(2, 2, 64, 64, seed=22)]
    vals = []
    for i, (xl, y, xu) in enumerate(batches):
        for tr in (eager, graph):
            uaps_amd.perturb.manual_seed(5, 0)
            np.random.seed(5)
            tr.train_step(xl, y, xu)
        if i in (3, 5):                               # after the capture (step 2) and between replays
            vals.append((eager.validate(val), graph.validate(val)))
    assert graph.step_graph.graph is not None
    for ve, vg in vals:
        assert ve == vg
    for (n, pa), pb in zip(m0.named_parameters(), m1.parameters()):
        assert torch.equal(pa, pb), n
    for (n, ba), bb in zip(m0.named_buffers(), m1.buffers()):
        assert torch.equal(ba, bb), n


@pytest.mark.parametrize("streams", [False, True])
def test_long_unsynchronised_replay_run_matches_the_eager_state_steps(streams, monkeypatch):
    """24 steps with no host synchronisation inside the loop (the host runs many replays ahead of the GPU: what bench.py and a
    training loop do): parameters, buffers and Adam state of the replayed run equal the eager state-mode run bit for bit."""
    import uaps_amd
    from uaps_amd import unet
    monkeypatch.setattr(unet, "_DECODER_STREAMS", streams)
    m0 = _model(12)
    m1 = copy.deepcopy(m0)
    m0.to(DEV), m1.to(DEV)
    data = _batches(4, 2, 64, 64, seed=31)
    runs = []
    for model, kw in ((m0, {"step_state": True}), (m1, {"use_graph": True})):
        tr = uaps_amd.UAPSTrainer(model, base_lr=1e-3, seed=9, **kw)
        uaps_amd.perturb.manual_seed(9, 0)
        np.random.seed(9)
        for i in range(24):
            tr.train_step(*data[i % 4])
        runs.append(tr)
    torch.cuda.synchronize()
    assert runs[1].step_graph.graph is not None
    for (n, pa), pb in zip(m0.named_parameters(), m1.parameters()):
        assert torch.equal(pa, pb), n
    for (n, ba), bb in zip(m0.named_buffers(), m1.buffers()):
        assert torch.equal(ba, bb), n
    assert float(runs[0].last["loss"]) == float(runs[1].last["loss"])
    assert runs[0].epoch_metrics() == runs[1].epoch_metrics()


@pytest.mark.parametrize("streams", [False, True])
def test_graph_replay_equals_the_eager_step_in_every_arithmetic(streams, monkeypatch):
    """The default arithmetic (two fp16 pieces per scaled operand, scales from device-resident bounds that the captured step
    zero-fills, raises and reads every replay) and the bf16 three-piece form: six steps replayed and eager give the same losses
    and parameters BIT FOR BIT (until round 3 this held only to 1e-5 / 1e-4 in these modes: DESIGN.md section 4)."""
    import uaps_amd
    from uaps_amd import conv, unet
    assert conv.get_mode() == "h16"
    monkeypatch.setattr(unet, "_DECODER_STREAMS", streams)
    try:
        for mode in ("h16", "split"):
            conv.set_mode(mode)
            m0 = _model(4)
            m1 = copy.deepcopy(m0)
            m0.to(DEV), m1.to(DEV)
            eager = uaps_amd.UAPSTrainer(m0, base_lr=1e-3, seed=7, step_state=True)
            graph = uaps_amd.UAPSTrainer(m1, base_lr=1e-3, seed=7, use_graph=True)
            le, lg = [], []
            for xl, y, xu in _batches(6, 2, 64, 64):
                for tr, acc in ((eager, le), (graph, lg)):
                    uaps_amd.perturb.manual_seed(7, 0)
                    np.random.seed(7)
                    acc.append(tr.train_step(xl, y, xu)["loss"].clone())
            torch.cuda.synchronize()
            assert graph.step_graph.graph is not None
            assert [float(v) for v in lg] == [float(v) for v in le], mode
            for (n, pa), pb in zip(m0.named_parameters(), m1.parameters()):
                assert torch.equal(pa, pb), (mode, n)
    finally:
        conv.set_mode("h16")


def test_graph_trainer_takes_a_ragged_batch_and_a_caller_supplied_mix():
    """A trainer built with use_graph=True whose step cannot go through the state-mode body (x_l.shape != x_u.shape: a ragged
    last batch; or caller-supplied mixing weights) runs the plain eager step for that call -- Adam's scalars by value -- and
    replays again afterwards with the right Adam step count."""
    import uaps_amd
    m0 = _model(6)
    m1 = copy.deepcopy(m0)
    m0.to(DEV), m1.to(DEV)
    plain = uaps_amd.UAPSTrainer(m0, base_lr=1e-3, seed=3)
    graph = uaps_amd.UAPSTrainer(m1, base_lr=1e-3, seed=3, use_graph=True)
    same = _batches(5, 2, 64, 64, seed=21)
    (xl, y, _), = _batches(1, 2, 64, 64, seed=22)
    (_, _, xu3), = _batches(1, 3, 64, 64, seed=23)
    seq = same[:3] + [(xl, y, xu3)] + same[3:]
    for i, (a, b, c) in enumerate(seq):
        r = graph.train_step(a, b, c)
        assert np.isfinite(float(r["loss"])), i
    assert graph.step_graph.graph is not None and graph.iter_num == 6
    steps = {float(st["step"]) for st in graph.optimizer.state.values()}
    assert steps == {6.0}
    # explicit mixing weights: eager route as well
    w = np.full(4, 0.25)
    r = graph.train_step(*same[0], w=w)
    assert np.isfinite(float(r["loss"])) and np.array_equal(r["w"], w)
    del plain


def test_returned_scalars_are_not_aliases_of_the_graph_outputs():
    import uaps_amd
    m = _model(8).to(DEV)
    tr = uaps_amd.UAPSTrainer(m, base_lr=1e-2, seed=2, use_graph=True)
    kept = [tr.train_step(*b)["loss"] for b in _batches(6, 2, 64, 64, seed=31)]
    vals = [float(v) for v in kept]
    assert tr.step_graph.graph is not None and len(set(vals[2:])) == 4, vals      # replayed steps keep their own values


def test_checkpoint_load_drops_the_capture(tmp_path):
    """load_checkpoint replaces Adam's moment tensors: a captured step would keep updating the old buffers through its frozen
    pointers.  After a load the step warms up and captures again, and continues exactly like an eager state-mode trainer that
    loaded the same checkpoint."""
    import os
    import uaps_amd
    if True:
        m0 = _model(9)
        m1, m2 = copy.deepcopy(m0), copy.deepcopy(m0)
        m0.to(DEV), m1.to(DEV), m2.to(DEV)
        bs = _batches(10, 2, 64, 64, seed=41)
        src = uaps_amd.UAPSTrainer(m0, base_lr=1e-3, seed=4, use_graph=True)
        for b in bs[:4]:
            src.train_step(*b)
        path = os.path.join(tmp_path, "ck.pth")
        src.save_checkpoint(path, 1, 0.5)
        assert "step_key" in torch.load(path, weights_only=False)
        g = uaps_amd.UAPSTrainer(m1, base_lr=1e-3, seed=4, use_graph=True)
        e = uaps_amd.UAPSTrainer(m2, base_lr=1e-3, seed=4, step_state=True)
        for b in bs[4:7]:                            # train a little first, so that a capture exists when the checkpoint arrives
            g.train_step(*b)
        assert g.step_graph.graph is not None
        g.load_checkpoint(path); e.load_checkpoint(path)
        assert g.step_graph.graph is None and g.step_graph.state.key == src.step_graph.state.key == e.step_graph.state.key
        for i, b in enumerate(bs[4:]):
            uaps_amd.perturb.manual_seed(4, 0); np.random.seed(4)
            le = float(e.train_step(*b)["loss"])
            uaps_amd.perturb.manual_seed(4, 0); np.random.seed(4)
            lg = float(g.train_step(*b)["loss"])
            assert le == lg, (i, le, lg)
        assert g.step_graph.graph is not None
        for (n, pa), pb in zip(m1.named_parameters(), m2.parameters()):
            assert torch.equal(pa, pb), n
        assert {float(st["step"]) for st in g.optimizer.state.values()} == {10.0}


def test_two_trainers_interleave_their_step_states():
    """The step-state pointer is process-wide (include/uaps_hip.h): every state-mode step brackets its launches with
    uaps_set_step_state(own) ... (NULL), so two trainers of one process (the two-model methods of the reference's siblings, CPS)
    can step alternately -- each ends exactly where it ends when it runs alone, one of them replaying a captured graph."""
    import uaps_amd
    data = _batches(6, 2, 64, 64, seed=51)

    def run(models, kws, seeds):
        trs = [uaps_amd.UAPSTrainer(m, base_lr=1e-3, seed=s, **kw) for m, kw, s in zip(models, kws, seeds)]
        for i in range(6):
            for tr, s in zip(trs, seeds):
                uaps_amd.perturb.manual_seed(s, 0); np.random.seed(s)
                tr.train_step(*data[(i + s) % 6])
        torch.cuda.synchronize()
        return trs

    base = [_model(13), _model(14)]
    solo = []
    for k, (kw, seed) in enumerate((({"use_graph": True}, 3), ({"step_state": True}, 4))):
        m = copy.deepcopy(base[k]).to(DEV)
        run([m], [kw], [seed])
        solo.append(m)
    both = [copy.deepcopy(base[0]).to(DEV), copy.deepcopy(base[1]).to(DEV)]
    trs = run(both, [{"use_graph": True}, {"step_state": True}], [3, 4])
    assert trs[0].step_graph.graph is not None
    for k in range(2):
        for (n, pa), pb in zip(solo[k].named_parameters(), both[k].parameters()):
            assert torch.equal(pa, pb), (k, n)
        for (n, ba), bb in zip(solo[k].named_buffers(), both[k].buffers()):
            assert torch.equal(ba, bb), (k, n)


@pytest.mark.parametrize("streams", [False, True])
def test_deferred_weight_gradient_reductions_give_the_same_step_bit_for_bit(streams, monkeypatch):
    """conv.deferred_reduces (the trainers' scope: one batched reduction launch behind the backward instead of one launch per
    convolution) against the immediate reductions: same gradients, hence the same parameters after three steps, bit for bit --
    eagerly, with decoder streams, and through the captured graph."""
    import uaps_amd
    from uaps_amd import conv, stepctx
    import uaps_amd.unet as _unet
    monkeypatch.setattr(_unet, "_DECODER_STREAMS", streams)
    data = _batches(3, 2, 64, 64)
    results = []
    for defer, graph in ((False, False), (True, False), (True, True)):
        monkeypatch.setattr(conv, "_DEFER", defer)
        m = _model(7).to(DEV)
        tr = uaps_amd.UAPSTrainer(m, base_lr=1e-3, seed=11, step_state=True, use_graph=graph)
        flushed = []
        orig = conv.flush_weight_reduces
        monkeypatch.setattr(conv, "flush_weight_reduces", lambda step=None: flushed.append(orig(step)) or flushed[-1])
        for xl, y, xu in data + data:
            tr.train_step(xl, y, xu)
        monkeypatch.setattr(conv, "flush_weight_reduces", orig)
        torch.cuda.synchronize()
        tr.check_errors()
        assert stepctx.current() is None                    # no scope left open on this thread
        if defer:
            assert flushed and all(n > 40 for n in flushed)   # every convolution of the net rode in the batch
        else:
            assert not any(flushed)
        results.append({n: p.detach().cpu().numpy() for n, p in m.named_parameters()})
    for other in results[1:]:
        for n, a in results[0].items():
            np.testing.assert_array_equal(other[n], a, err_msg=n)


def test_a_failing_backward_drops_the_pending_reductions():
    import uaps_amd
    from uaps_amd import conv, stepctx
    x = torch.randn(2, 8, 32, 32, device=DEV, requires_grad=True)
    w = torch.randn(16, 8, 3, 3, device=DEV, requires_grad=True)
    with pytest.raises(RuntimeError, match="boom"):
        with conv.deferred_reduces() as step:
            y = conv.conv2d(x, w, None)
            y.sum().backward()
            assert step is stepctx.current() and step.deferred and len(step.deferred) == 1
            raise RuntimeError("boom")
    assert stepctx.current() is None and step.deferred is None
    # outside a scope the reduction is immediate and the gradient is the oracle's
    w.grad = None
    conv.conv2d(x, w, None).sum().backward()
    ref = torch.nn.grad.conv2d_weight(x.detach().cpu().double(), w.shape, torch.ones(2, 16, 32, 32, dtype=torch.float64), padding=1)
    np.testing.assert_allclose(w.grad.cpu().numpy(), ref.numpy(), rtol=2e-5, atol=2e-4)


def test_a_parameter_with_a_foreign_gradient_hook_or_a_derived_weight_is_reduced_at_once():
    """Deferral is for leaf parameters nobody watches: a post-accumulate hook of the caller's reads the true gradient inside the
    backward, and the gradient of a non-leaf weight (read by the next autograd node) is complete when it is handed over."""
    from uaps_amd import conv
    torch.manual_seed(2)
    x = torch.randn(2, 8, 32, 32, device=DEV)
    w = torch.randn(16, 8, 3, 3, device=DEV, requires_grad=True)
    ref = torch.nn.grad.conv2d_weight(x.cpu().double(), w.shape, torch.ones(2, 16, 32, 32, dtype=torch.float64), padding=1).numpy()
    seen = []
    w.register_post_accumulate_grad_hook(lambda p: seen.append(p.grad.detach().clone()))
    with conv.deferred_reduces() as step:
        conv.conv2d(x, w, None).sum().backward()
        assert not step.deferred
    np.testing.assert_allclose(seen[0].cpu().numpy(), ref, rtol=2e-5, atol=2e-4)
    v = torch.randn(16, 8, 3, 3, device=DEV, requires_grad=True)
    with conv.deferred_reduces() as step:
        conv.conv2d(x, v * 2.0, None).sum().backward()          # the weight is a non-leaf: its gradient feeds the multiplication's backward
        assert not step.deferred
    np.testing.assert_allclose(v.grad.cpu().numpy(), 2.0 * ref, rtol=2e-5, atol=4e-4)


def test_a_replayed_step_launches_no_library_kernel():
    """with the batch written into the graph's own input tensors (StepGraph.inputs()) and the live output scalars
    (StepGraph.live_outputs), a replayed training step is hand-written kernels plus ONE 64-byte upload of the step state (the host-drawn
    Dirichlet weights, ramp weights, Adam scalars and Philox key: graph.StepState.upload): no ATen elementwise / fill / reduction
    kernel and no other runtime copy appears in a profiled replay.  (The default -- caller-owned inputs, cloned output scalars -- adds
    three input copies and three scalar clones OUTSIDE the graph: tools/diag/replay_library_launches.py lists them.)"""
    import uaps_amd
    from torch.profiler import ProfilerActivity, profile
    tr = uaps_amd.UAPSTrainer(_model(9).to(DEV), base_lr=1e-3, seed=0, use_graph=True, track_metrics=False)
    data = _batches(1, 4, 64, 64)[0]
    for _ in range(4):                           # warm-up steps, the capture, one replay
        tr.train_step(*data)
    g = tr.step_graph
    assert g.graph is not None and g.inputs() is not None
    g.live_outputs = True
    xs = g.inputs()
    for dst, src in zip(xs, data):               # the "data pipeline": writes the batch where the graph reads it
        dst.copy_(src)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        last = tr.train_step(*xs)
        torch.cuda.synchronize()
    dev_names = [str(e.name) for e in prof.events() if e.device_type.name == "CUDA"]
    own = [n for n in dev_names if "uaps::" in n or "(anonymous namespace)::" in n]
    other = [n for n in dev_names if n not in own]
    assert len(own) > 100, len(own)              # the graph's kernels are in the trace
    assert not [n for n in other if not n.startswith("Memcpy")], other
    assert len(other) <= 1, other                # the step-state upload
    assert last["loss"].data_ptr() == g.static["out"].loss.data_ptr()      # the live tensor, not a copy
    assert bool(torch.isfinite(last["loss"]))
