"""On-device input pipeline (csrc/augment.hip, uaps_amd/augment.py) against its numpy restatement
(oracle/augment_oracle.py).  Parity with the reference's cv2 + albumentations loader is UNPINNED (neither library is in
the image): these tests pin the kernel to the stage-by-stage restatement only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _batch(rng, B, Hs, Ws, C):
    img = rng.integers(0, 256, (B, Hs, Ws, 3), dtype=np.uint8)
    mask = rng.integers(0, C, (B, Hs, Ws), dtype=np.uint8)
    return img, mask


@pytest.mark.parametrize("Hs,Ws,Ho,Wo", [(200, 200, 256, 256), (256, 256, 256, 256), (96, 160, 64, 64), (37, 53, 32, 32)])
def test_augment_matches_numpy_restatement(Hs, Ws, Ho, Wo):
    from uaps_amd import augment
    from oracle import augment_oracle as AO
    rng = np.random.default_rng(Hs * 7 + Wo)
    B = 12
    img, mask = _batch(rng, B, Hs, Ws, 4)
    params = augment.draw_train_params(B, rng)
    # make sure every stage and every combination end is exercised at least once
    params.ints[0, :5] = (1, 1, 1, 7, 1); params.floats[0, :3] = (1.5, 0.5, 7.0)
    params.ints[1, :5] = (0, 0, 3, 3, 0); params.floats[1, :3] = (1.0, 0.0, 0.0)
    params.ints[2, :5] = (1, 0, 2, 5, 1); params.floats[2, :3] = (1.2, 0.1, 3.3)
    params.ints[3, :5] = (0, 0, 0, 0, 0); params.floats[3, :3] = (1.0, 0.0, 0.0)
    noise = (rng.standard_normal((B, 3, Ho, Wo)).astype(np.float32) * params.floats[:, 2][:, None, None, None]).astype(np.float32)
    x, y = augment.augment_batch(torch.from_numpy(img).to(DEV), torch.from_numpy(mask).to(DEV), params, (Ho, Wo),
                                 noise=torch.from_numpy(noise).to(DEV))
    x, y = x.cpu().numpy(), y.cpu().numpy()
    for b in range(B):
        rx, ry = AO.augment_one(img[b], mask[b], params.ints[b], params.floats[b], noise[b], Ho, Wo, augment.IMAGENET_MEAN,
                                augment.IMAGENET_STD)
        assert np.array_equal(y[b], ry), f"mask {b}"
        # grey levels are integers: a mismatch would be >= 1/255/std = 0.017; allow float rounding of the normalisation only
        bad = np.abs(x[b] - rx) > 1e-5
        assert bad.mean() < 2e-4, f"image {b}: {bad.sum()} of {bad.size} pixels differ"     # ties of v*alpha+beta*255 / x.5 means


def test_identity_params_is_resize_and_normalise():
    from uaps_amd import augment
    rng = np.random.default_rng(3)
    img, mask = _batch(rng, 4, 200, 200, 4)
    x, y = augment.augment_batch(torch.from_numpy(img).to(DEV), torch.from_numpy(mask).to(DEV), augment.identity_params(4))
    ys = (np.arange(256) * 200) // 256
    ref = img[:, ys][:, :, ys].astype(np.float32) / 255.0
    ref = (ref - np.asarray(augment.IMAGENET_MEAN, np.float32)) / np.asarray(augment.IMAGENET_STD, np.float32)
    np.testing.assert_allclose(x.cpu().numpy(), np.transpose(ref, (0, 3, 1, 2)), atol=1e-5)
    assert np.array_equal(y.cpu().numpy(), mask[:, ys][:, :, ys].astype(np.int64))


def test_in_kernel_noise_statistics_and_train_step_consumes_the_batch():
    """Philox / Box-Muller path: mean ~ 0 and std ~ sigma on a mid-grey image (no clipping); the produced batch feeds a
    training step unchanged."""
    import uaps_amd
    from uaps_amd import augment
    B = 4
    img = np.full((B, 64, 64, 3), 128, np.uint8)
    mask = np.zeros((B, 64, 64), np.uint8); mask[:, 10:30, 20:50] = 2
    p = augment.identity_params(B)
    p.ints[:, 4] = 1; p.floats[:, 2] = 6.0
    x, y = augment.augment_batch(torch.from_numpy(img).to(DEV), torch.from_numpy(mask).to(DEV), p, (64, 64), seed=1234)
    grey = (x.cpu().numpy() * np.asarray(augment.IMAGENET_STD, np.float32)[None, :, None, None]
            + np.asarray(augment.IMAGENET_MEAN, np.float32)[None, :, None, None]) * 255.0
    d = grey - 128.0
    assert abs(d.mean() + 0.5) < 0.15          # truncation toward zero of v + g biases the mean by about -0.5 grey levels
    assert abs(d.std() - 6.0) < 0.3
    x2, _ = augment.augment_batch(torch.from_numpy(img).to(DEV), None, p, (64, 64), seed=1234)
    assert torch.equal(x, x2)                   # same seed, same draw
    model = uaps_amd.net_factory("unet_uaps", 3, 4).to(DEV)
    tr = uaps_amd.UAPSTrainer(model)
    out = tr.train_step(x, y, x.flip(0))
    assert torch.isfinite(out["loss"]).item()
