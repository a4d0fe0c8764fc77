"""-m gpu: a violated magnitude bound is REPORTED.  The fp16-split convolutions (conv mode 2) scale their operands by a power
of two derived from a device-resident upper bound; a bound that is too small overflows fp16 and the outputs are NaN.  The
kernels check what they store and raise a bit of the sticky device error word (include/uaps_hip.h, uaps_set_error_word), and
the trainer reads it with the device->host copies it makes anyway and raises."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _word():
    """The package's never-freed word of the device (uaps_amd._lib.error_word), zeroed."""
    from uaps_amd import _lib
    w = _lib.error_word(torch.device(DEV))
    w.zero_()
    return w


@pytest.mark.parametrize("shape", [(2, 32, 32, 64), (2, 16, 16, 64), (2, 64, 64, 32)])
def test_a_bound_that_is_too_small_sets_the_error_word(shape):
    from uaps_amd import _lib, bounds, conv
    assert conv.get_mode() == "h16"
    B, Cin, Cout, HW = shape
    torch.manual_seed(0)
    x = torch.randn(B, Cin, HW, HW, device=DEV)
    w = torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.05
    dy = torch.randn(B, Cout, HW, HW, device=DEV)
    wf, wb = conv.pack_weights(w)
    amax_x, amax_dy = x.abs().max(), dy.abs().max()
    word = _word()
    good_x, good_dy = (bounds.from_value(amax_x), 1.0), (bounds.from_value(amax_dy), 1.0)
    y = conv.conv_fwd_raw(x, wf, None, Cout, 3, 0, xb=good_x)
    conv.conv_bwd_data_raw(dy, wb, Cin, 3, 0, dyb=good_dy)
    conv.conv_bwd_weight_raw(dy, x, 3, True, 0, dyb=good_dy, xb=good_x)
    assert int(word.item()) == 0 and bool(torch.isfinite(y).all())
    # the scale leaves 2x of headroom: a bound 1.5x too small is still exact
    y15 = conv.conv_fwd_raw(x, wf, None, Cout, 3, 0, xb=(bounds.from_value(amax_x / 1.5), 1.0))
    assert int(word.item()) == 0 and torch.equal(y15, y)
    # 100x too small: overflow, NaN outputs, reported
    bad_x, bad_dy = (bounds.from_value(amax_x / 100), 1.0), (bounds.from_value(amax_dy / 100), 1.0)
    yb = conv.conv_fwd_raw(x, wf, None, Cout, 3, 0, xb=bad_x)
    assert int(word.item()) & 1 and not bool(torch.isfinite(yb).all())
    word.zero_()
    conv.conv_bwd_data_raw(dy, wb, Cin, 3, 0, dyb=bad_dy)
    assert int(word.item()) & 1
    word.zero_()
    conv.conv_bwd_weight_raw(dy, x, 3, True, 0, dyb=good_dy, xb=bad_x)
    assert int(word.item()) & 2
    word.zero_()
    # non-finite data with a true bound is reported as well (the message of UAPSTrainer.check_errors names both causes)
    xn = x.clone(); xn[0, 0, 3, 3] = float("inf")
    conv.conv_fwd_raw(xn, wf, None, Cout, 3, 0, xb=good_x)
    assert int(word.item()) & 1
    word.zero_()


def test_trainer_raises_when_a_bound_is_violated(monkeypatch):
    import uaps_amd
    from uaps_amd import _lib, bounds
    torch.manual_seed(1)
    model = uaps_amd.UNet_UAPS(3, 4, feature_chns=[16, 16, 32, 32, 64]).to(DEV)
    tr = uaps_amd.UAPSTrainer(model, seed=3)
    rng = np.random.default_rng(0)
    xl = torch.tensor(rng.standard_normal((2, 3, 64, 64)).astype(np.float32)).to(DEV)
    xu = torch.tensor(rng.standard_normal((2, 3, 64, 64)).astype(np.float32)).to(DEV)
    y = torch.tensor(uaps_amd.data.synthetic_masks(rng, 2, 4, 64, 64)).to(DEV)
    tr.train_step(xl, y, xu)
    assert np.isfinite(tr.epoch_metrics()["miou"])            # a healthy step: nothing is reported
    real = bounds.bn_output_bound
    monkeypatch.setattr(bounds, "bn_output_bound", lambda bn, n, factor=1.0: (lambda b: None if b is None else (b[0], b[1] * 1e-5))(real(bn, n, factor)))
    tr.train_step(xl, y, xu)                                   # every BatchNorm output bound 1e5 x too small
    with pytest.raises(_lib.UapsHipError, match="magnitude bound"):
        tr.epoch_metrics()
    monkeypatch.undo()
    assert int(tr._err.item()) == 0                            # the word is sticky until read, then cleared


def test_the_error_word_outlives_the_trainer_that_bound_it():
    """The library keeps a raw pointer to the error word: it must not dangle once a trainer is dropped (bench.py's other_configs
    drops trainers and empties the allocator's cache).  The word is per device and never freed; a convolution that overflows
    after the trainer is gone reports into it, and the next trainer on the device reads it."""
    import gc
    import uaps_amd
    from uaps_amd import _lib, bounds, conv
    torch.manual_seed(2)
    model = uaps_amd.UNet_UAPS(3, 4, feature_chns=[16, 16, 32, 32, 64]).to(DEV)
    tr = uaps_amd.UAPSTrainer(model, seed=3)
    ptr = tr._err.data_ptr()
    del tr, model
    gc.collect()
    torch.cuda.empty_cache()
    filler = [torch.zeros(1 << 20, device=DEV) for _ in range(8)]      # whatever the allocator hands out next
    x = torch.randn(2, 32, 32, 32, device=DEV)
    w = torch.randn(32, 32, 3, 3, device=DEV) * 0.05
    wf, _ = conv.pack_weights(w)
    conv.conv_fwd_raw(x, wf, None, 32, 3, 0, xb=(bounds.from_value(x.abs().max() / 100), 1.0))
    torch.cuda.synchronize()
    assert all(float(f.abs().max()) == 0.0 for f in filler)
    tr2 = uaps_amd.UAPSTrainer(uaps_amd.UNet_UAPS(3, 4, feature_chns=[16, 16, 32, 32, 64]).to(DEV), seed=3)
    assert tr2._err.data_ptr() == ptr
    with pytest.raises(_lib.UapsHipError, match="magnitude bound"):
        tr2.check_errors()
    tr2.check_errors()                                                  # read and cleared
