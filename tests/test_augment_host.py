"""CPU checks of the input pipeline's host side (uaps_amd/augment.py parameter draws) and of the numpy restatement the
GPU kernel is tested against (oracle/augment_oracle.py).  Parity with cv2 / albumentations is unpinned."""
import numpy as np

from oracle import augment_oracle as AO
from uaps_amd import augment


def test_train_parameter_draws_follow_the_loader_probabilities():
    p = augment.draw_train_params(20000, np.random.default_rng(0))
    i, f = p.ints, p.floats
    assert abs(i[:, 0].mean() - 0.4) < 0.02 and abs(i[:, 1].mean() - 0.4) < 0.02           # flips p=0.4 (dataloaders.py:98)
    assert abs((i[:, 3] > 0).mean() - 0.3) < 0.02 and set(np.unique(i[:, 3])) == {0, 3, 5, 7}   # Blur p=0.3, k odd 3..7
    assert abs((f[:, 0] != 1.0).mean() - 0.5) < 0.02                                       # RandomBrightnessContrast p=0.5
    assert f[:, 0].min() >= 1.0 and f[:, 0].max() <= 1.5 and f[:, 1].min() >= 0.0 and f[:, 1].max() <= 0.5
    assert abs(i[:, 4].mean() - 0.3) < 0.02                                                # GaussNoise p=0.3
    s = f[i[:, 4] == 1, 2]
    assert s.min() >= np.sqrt(10.0) - 1e-6 and s.max() <= np.sqrt(50.0) + 1e-6             # var_limit (10, 50)
    assert (f[i[:, 4] == 0, 2] == 0).all()
    rot = i[:, 2]
    assert abs((rot > 0).mean() - 0.3 * 0.75) < 0.02                                       # RandomRotate90 p=0.3, k uniform in 0..3


def test_numpy_restatement_stage_properties():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    mask = rng.integers(0, 4, (40, 56), dtype=np.uint8)
    ident = augment.identity_params(1)
    zero = np.zeros((3, 32, 32), np.float32)
    x, m = AO.augment_one(img, mask, ident.ints[0], ident.floats[0], zero, 32, 32, (0, 0, 0), (1, 1, 1))
    ys, xs = (np.arange(32) * 40) // 32, (np.arange(32) * 56) // 32
    assert np.array_equal(m, mask[ys][:, xs]) and np.allclose(x, np.transpose(img[ys][:, xs], (2, 0, 1)) / 255.0)
    # four quarter turns, two flips: back to the start
    ints = ident.ints[0].copy(); ints[0] = 1
    x1, m1 = AO.augment_one(img, mask, ints, ident.floats[0], zero, 32, 32, (0, 0, 0), (1, 1, 1))
    assert np.array_equal(m1, m[:, ::-1]) and np.allclose(x1, x[:, :, ::-1])
    ints = ident.ints[0].copy(); ints[2] = 1
    x2, m2 = AO.augment_one(img, mask, ints, ident.floats[0], zero, 32, 32, (0, 0, 0), (1, 1, 1))
    assert np.array_equal(m2, np.rot90(m, 1))
    # the box blur of a constant image is the constant; brightness saturates at 255
    const = np.full((8, 8, 3), 77, np.uint8)
    assert (AO.box_blur(const, 5) == 77).all()
    assert (AO.brightness_contrast(const, 1.5, 0.5) == min(255, int(77 * 1.5 + 127.5))).all()
    assert (AO.brightness_contrast(np.full((2, 2, 3), 250, np.uint8), 1.5, 0.5) == 255).all()
