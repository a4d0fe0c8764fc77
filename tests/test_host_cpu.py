"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/uaps_hip.h
declares, the model's checkpoint layout equals the reference's, host logic (ramp, metrics from a
confusion matrix, checkpoint round trip) matches the fixtures, and the product ops refuse to run
without a GPU instead of falling back."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT


def test_library_exports_every_declared_symbol():
    from uaps_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "uaps_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(uaps_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    assert os.path.exists(_lib.LIB_PATH), "libuaps_hip.so not built: run python -c 'import __graft_entry__ as g; g.build()'"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in uaps_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.lib()
    assert L.uaps_abi_version() == 3
    assert b"range" in L.uaps_error_string(-2) and b"no form" in L.uaps_error_string(-4)


def test_argument_validation_without_gpu():
    """Calls that return before any launch: bad dimensions / null pointers / workspace queries."""
    from uaps_amd import _lib
    L = _lib.lib()
    n = ctypes.c_size_t()
    assert L.uaps_loss_workspace_bytes(4, 16, 4, 256, 256, ctypes.byref(n)) == 0 and n.value >= 1024 * 48 * 4
    assert L.uaps_loss_workspace_bytes(9, 1, 4, 8, 8, ctypes.byref(n)) == -2
    assert L.uaps_loss_workspace_bytes(4, 1, 1, 8, 8, ctypes.byref(n)) == -2
    assert L.uaps_loss_workspace_bytes(4, 0, 4, 8, 8, ctypes.byref(n)) == -1
    assert L.uaps_unsup_fwd(None, None, 4, 1, 4, 8, 8, 0.1, 0.1, 1e-7, None, None, None, None, 0, None) == -1
    assert L.uaps_feat_dropout_workspace_bytes(2, 4, 8, 8, ctypes.byref(n)) == 0 and n.value >= 2 * 64 * 4
    assert L.uaps_seg_confusion(None, None, 1, 4, 8, 8, None, None) == -1


def test_state_dict_layout_equals_reference():
    import uaps_amd
    g = np.load(os.path.join(GOLDEN, "g4_model.npz"))
    for in_chns, ncls in ((3, 4), (1, 7), (3, 2), (3, 6)):
        net = uaps_amd.UNet_UAPS(in_chns, ncls)
        sd = net.state_dict()
        assert list(sd.keys()) == list(g[f"keys_{in_chns}_{ncls}"])
        assert [",".join(map(str, v.shape)) for v in sd.values()] == list(g[f"shapes_{in_chns}_{ncls}"])
        assert [str(v.dtype) for v in sd.values()] == list(g[f"dtypes_{in_chns}_{ncls}"])
        assert [n for n, _ in net.named_parameters()] == list(g[f"param_names_{in_chns}_{ncls}"])
        assert sum(p.numel() for p in net.parameters()) == int(g[f"nparams_{in_chns}_{ncls}"])
    assert list(uaps_amd.UNet(3, 4).state_dict().keys()) == list(g["keys_unet_3_4"])
    assert sum(p.numel() for p in uaps_amd.UNet_UAPS(3, 4).parameters()) == 3713952
    # K=5 stress config: two more decoders with the same layout
    k5 = uaps_amd.UNet_UAPS(1, 2, n_aux=5)
    assert any(k.startswith("aux_decoder5.up1.conv1x1") for k in k5.state_dict())


def test_net_factory_surface():
    import uaps_amd
    assert uaps_amd.net_factory("nope") is None
    m = uaps_amd.net_factory("unet_uaps", in_chns=3, class_num=4)
    assert isinstance(m, uaps_amd.UNet_UAPS) and m.n_aux == 3
    assert isinstance(uaps_amd.net_factory("unet"), uaps_amd.UNet)
    with pytest.raises(AssertionError):
        uaps_amd.unet.Encoder(3, feature_chns=[1, 2, 3])


def test_ramp_matches_reference():
    import uaps_amd
    g = np.load(os.path.join(GOLDEN, "g2_losses.npz"))
    for i, t in enumerate(g["ramp_t"]):
        for j, R in enumerate(g["ramp_R"]):
            assert abs(uaps_amd.sigmoid_rampup(t, R) - g["ramp"][i, j]) < 1e-12
    assert uaps_amd.get_current_consistency_weight(0.1, 159) == 0.1 * uaps_amd.sigmoid_rampup(1, 200)


def test_metrics_host_math_matches_reference():
    import uaps_amd
    from oracle import c_oracle
    g = np.load(os.path.join(GOLDEN, "g5_metrics.npz"))
    for i in range(4):
        cm = c_oracle.confusion(g[f"logits{i}"], g[f"labels{i}"])
        m = uaps_amd.metrics_from_confusion(cm)
        for key in ("miou", "mdice", "acc"):
            ref = float(g[f"{key}{i}"])
            assert (np.isnan(ref) and np.isnan(m[key])) or abs(m[key] - ref) < 1e-12


def test_model_modules_match_reference_blocks_on_cpu():
    """The nn.Module tree (torch ops, runs anywhere) against the reference blocks of fixture g4."""
    from uaps_amd import unet
    g = np.load(os.path.join(GOLDEN, "g4_model.npz"))
    cb = unet.ConvBlock(3, 4, 0.0)
    cb.load_state_dict({k[3:]: torch.tensor(g[k]) for k in g.files if k.startswith("cb.")})
    x = torch.tensor(g["cb_x"])
    cb.train(); np.testing.assert_allclose(cb(x).detach().numpy(), g["cb_y_train"], atol=2e-6)
    for k, v in cb.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g["cb_after." + k], atol=1e-6)
    cb.eval(); np.testing.assert_allclose(cb(x).detach().numpy(), g["cb_y_eval"], atol=2e-6)
    ub = unet.UpBlock(8, 4, 4)
    ub.load_state_dict({k[3:]: torch.tensor(g[k]) for k in g.files if k.startswith("ub.")})
    ub.train(); np.testing.assert_allclose(ub(torch.tensor(g["ub_x1"]), torch.tensor(g["ub_x2"])).detach().numpy(), g["ub_y_train"], atol=2e-6)
    ub.eval(); np.testing.assert_allclose(ub(torch.tensor(g["ub_x1"]), torch.tensor(g["ub_x2"])).detach().numpy(), g["ub_y_eval"], atol=2e-6)
    db = unet.DownBlock(4, 6, 0.0)
    db.load_state_dict({k[3:]: torch.tensor(g[k]) for k in g.files if k.startswith("db.")})
    db.train(); np.testing.assert_allclose(db(torch.tensor(g["db_x"])).detach().numpy(), g["db_y_train"], atol=2e-6)
    f = [2, 4, 8, 16, 32]
    enc, dec = unet.Encoder(3, f), unet.Decoder(4, f)
    enc.load_state_dict({k[len("narrow.encoder."):]: torch.tensor(g[k]) for k in g.files if k.startswith("narrow.encoder.")})
    dec.load_state_dict({k[len("narrow.main_decoder."):]: torch.tensor(g[k]) for k in g.files if k.startswith("narrow.main_decoder.")})
    enc.eval(); dec.eval()
    with torch.no_grad():
        np.testing.assert_allclose(dec(enc(torch.tensor(g["narrow_x"]))).numpy(), g["narrow_y_eval"], atol=1e-5)


def test_product_ops_refuse_cpu_tensors():
    import uaps_amd
    from uaps_amd._lib import UapsHipError
    z = [torch.randn(1, 4, 8, 8) for _ in range(4)]
    with pytest.raises(UapsHipError):
        uaps_amd.uaps_unsup_loss(z, [0.25] * 4, 0.1, 0.1)
    with pytest.raises(UapsHipError):
        uaps_amd.dice_loss(torch.zeros(1, 1, 8, 8, dtype=torch.long), z[0])
    with pytest.raises(UapsHipError):
        uaps_amd.FeatureNoise()(z[0])
    with pytest.raises(UapsHipError):
        uaps_amd.Dropout(z[0])
    with pytest.raises(UapsHipError):
        uaps_amd.FeatureDropout(z[0])
    with pytest.raises(UapsHipError):
        uaps_amd.mIoU(z[0], torch.zeros(1, 8, 8, dtype=torch.long))
    with pytest.raises(UapsHipError):
        uaps_amd.UNet_UAPS(3, 4)(torch.randn(2, 3, 32, 32))      # aux decoders need the HIP perturbations


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from uaps_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.UapsHipError, match="not built"):
        _lib.lib()


def test_checkpoint_layout_round_trip(tmp_path):
    """UAPS_train.py:443-450 dict, `module.`-prefixed keys as saved from nn.DataParallel (UAPS_model.py:13)."""
    import uaps_amd
    from uaps_amd.trainer import load_state_dict_any_prefix
    from oracle import uaps_oracle as O
    torch.manual_seed(0)
    net = uaps_amd.UNet_UAPS(3, 4)
    tr = uaps_amd.UAPSTrainer(net, loss_fn=lambda *a: None)
    path = str(tmp_path / "Checkpoints" / "UAPS_NEU_10P.pth")
    tr.iter_num = 1234
    tr.scheduler.step(0.4)
    tr.save_checkpoint(path, epoch=7, best_dice=0.5)
    ck = torch.load(path)              # the plain call of the reference's loaders and of user tools: weights_only=True since torch 2.6
    # the reference's four keys (UAPS_train.py:443-448) + the resume extras its loaders ignore
    assert {"epoch", "best_dice_1", "state_dict", "optimizer"} <= set(ck) <= {"epoch", "best_dice_1", "state_dict", "optimizer", "iter_num", "scheduler", "mix_rng", "step_key"}
    assert all(k.startswith("module.") for k in ck["state_dict"]) and len(ck["state_dict"]) == 334
    assert "module.encoder.in_conv.conv_conv.0.weight" in ck["state_dict"]
    # the reference's way of consuming it: DataParallel(model).load_state_dict(ckpt['state_dict'])
    net2 = torch.nn.DataParallel(uaps_amd.UNet_UAPS(3, 4))
    net2.load_state_dict(ck["state_dict"])
    net3 = uaps_amd.UNet_UAPS(3, 4)
    load_state_dict_any_prefix(net3, ck["state_dict"])
    for (k, a), (_, b) in zip(net.state_dict().items(), net3.state_dict().items()):
        assert torch.equal(a, b), k
    tr3 = uaps_amd.UAPSTrainer(net3, loss_fn=lambda *a: None)
    assert tr3.load_checkpoint(path)["epoch"] == 7
    assert tr3.iter_num == 1234 and tr3.scheduler.best == 0.4              # the ramp and the plateau state resume
    assert abs(tr3.consistency_weights()[0] - tr.consistency_weights()[0]) < 1e-15
    assert np.array_equal(tr3.mix_rng.dirichlet(np.ones(4), size=3), tr.mix_rng.dirichlet(np.ones(4), size=3))      # the mixing stream resumes
    # scheduler surface of UAPS_train.py:113, 402
    tr.scheduler.step(0.3)
    assert tr.optimizer.param_groups[0]["lr"] == 1e-3


def test_synthetic_batches_shapes():
    import uaps_amd
    d = uaps_amd.data.SyntheticBatches(3, 3, 4, 32, 32, n_batches=2, device="cpu")
    xl, yl, xu = d.next()
    assert xl.shape == (3, 3, 32, 32) and xu.shape == xl.shape and yl.shape == (3, 32, 32)
    assert yl.dtype == torch.int64 and 0 <= int(yl.min()) and int(yl.max()) <= 3
    frac = float((yl > 0).float().mean())
    assert 0.01 < frac < 0.5


def test_mean_batch_metrics_follow_the_reference_loop():
    """UAPS_train.py:388-399 averages per-batch mIoU / mDice (NaN-mean over the classes present in each batch); pooling the
    confusion matrices over the batches is a different number as soon as a class is missing from one batch."""
    import uaps_amd
    from oracle import uaps_oracle as O
    rng = np.random.default_rng(5)
    cms = []
    for i in range(4):
        lg = torch.tensor(rng.standard_normal((2, 4, 16, 16)).astype(np.float32))
        y = torch.tensor(rng.integers(0, 4, (2, 16, 16)))
        if i % 2:
            y[y == 3] = 1                                   # class 3 absent from every other batch
        cms.append(O.confusion(lg, y, 4).numpy())
    cms = np.stack(cms)
    m = uaps_amd.mean_batch_metrics(cms)
    per = [O.metrics_from_confusion(c) for c in cms]
    for k in ("miou", "mdice", "acc"):
        assert abs(m[k] - np.mean([p[k] for p in per])) < 1e-12
    pooled = uaps_amd.metrics_from_confusion(cms.sum(0))
    assert abs(pooled["mdice"] - m["mdice"]) > 1e-6      # the two conventions differ on this data


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the failure path of a box without a GPU")
def test_bench_starts_its_own_ranks_and_propagates_their_failure():
    """`python bench.py --gpus 2` without a launcher starts two rank processes itself (bench.spawn_ranks) and exits with their
    code: without a GPU both ranks fail at the first device call, and the command fails with them instead of hanging."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    # (the first rank to fail ends the job: the other one is terminated, with or without having reported itself)
    assert "No HIP GPUs are available" in r.stderr or "Traceback" in r.stderr, r.stderr[-2000:]


def test_the_driver_build_hook_runs_against_the_built_library():
    """__graft_entry__.build() is what the driver runs as its build check: it must accept the library as built (an ABI-version bump
    once left its assertion behind)."""
    import __graft_entry__ as g
    g.build()


def test_a_model_behind_a_foreign_gradient_reducer_is_not_deferred():
    """torch's DistributedDataParallel hangs its reducer on the AccumulateGrad nodes, where no Python-side check of
    the parameter can see it; a trainer therefore refuses to defer for such a model as a whole."""
    import torch.nn as nn
    from uaps_amd import conv

    class FakeDDP(nn.parallel.DistributedDataParallel):      # the class test is what matters; no process group is needed for it
        def __init__(self, module):
            nn.Module.__init__(self)
            self.module = module

    m = nn.Conv2d(3, 8, 3)
    assert conv.defer_allowed(m) and conv.defer_allowed(nn.DataParallel(m))
    assert not conv.defer_allowed(FakeDDP(m))
    with conv.deferred_reduces(model=FakeDDP(m)) as step:
        assert step is None
    with conv.deferred_reduces(model=m) as step:
        assert step is not None


def test_bench_rank_plan_under_a_16_cpu_quota_and_8_ranks():
    """bench.py's choice of CPU affinity and launch mode per rank: the driver's GPU boxes give a container 16
    CPUs; eight ranks then have two cores each -- no pinning (fewer than 4 per rank), eager launches (two cores are the floor the
    eager step was measured with); with 8 CPUs the two-graph form is chosen by itself; UAPS_GRAPH_MULTI forces either."""
    sys.path.insert(0, ROOT)
    import bench
    cpus16 = list(range(100, 116))                                  # a cgroup's cores need not start at 0
    for r in range(8):
        aff, per, gm = bench.plan_ranks(cpus16, 8, r)
        assert aff is None and per == 2 and gm is False
    assert bench.plan_ranks(cpus16, 8, 0, "1")[2] is True and bench.plan_ranks(list(range(8)), 8, 0, "0")[2] is False
    for r in range(8):
        aff, per, gm = bench.plan_ranks(list(range(8)), 8, r)      # one core per rank: launch-bound eagerly -> the captured form
        assert aff is None and per == 1 and gm is True
    seen = []
    for r in range(4):                                              # 16 CPUs, 4 ranks: disjoint slices of 4, pinned
        aff, per, gm = bench.plan_ranks(cpus16, 4, r)
        assert per == 4 and len(aff) == 4 and gm is False
        seen += aff
    assert sorted(seen) == cpus16
    assert bench.plan_ranks(cpus16, 1, 0) == (None, 16, False)       # one rank: nothing to share
    aff, per, gm = bench.plan_ranks(range(256), 8, 7)               # a whole 2 x 64-core host
    assert aff == list(range(224, 256)) and per == 32 and gm is False


def test_staleness_stamps_belong_to_the_owning_optimizer():
    """conv.stamp(parameter): what a packed copy / BatchNorm bound is valid for.  One optimizer's step must not invalidate another
    model's cached derivatives (round 6: a process-wide counter did, from another thread, in the middle of a forward), a manual
    invalidate_packed_weights() invalidates everything, and a parameter two optimizers share follows both."""
    import torch.nn as nn
    from uaps_amd import conv
    a, b = nn.Conv2d(3, 4, 3), nn.Conv2d(3, 4, 3)
    oa, ob = torch.optim.SGD(a.parameters(), lr=0.1), torch.optim.Adam(b.parameters(), lr=0.1)
    for m in (a, b):
        m(torch.randn(1, 3, 8, 8)).sum().backward()
    sa0, sb0 = conv.stamp(a.weight), conv.stamp(b.weight)
    oa.step()
    assert conv.stamp(a.weight) != sa0 and conv.stamp(a.bias) == conv.stamp(a.weight)
    assert conv.stamp(b.weight) == sb0                                   # b's caches stay valid while a trains
    sa1 = conv.stamp(a.weight)
    ob.step(); ob.step()
    assert conv.stamp(a.weight) == sa1 and conv.stamp(b.weight) != sb0
    conv.invalidate_packed_weights()
    assert conv.stamp(a.weight) != sa1                                   # the manual switch reaches everybody
    shared = torch.optim.SGD([a.weight], lr=0.1)                         # a second owner of a.weight
    s2 = conv.stamp(a.weight)
    shared.step()
    s3 = conv.stamp(a.weight)
    oa.step()
    assert s3 != s2 and conv.stamp(a.weight) != s3                       # both owners' steps are seen
