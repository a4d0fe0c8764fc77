"""Host side of the on-device input pipeline (csrc/augment.hip): draws the per-image
augmentation parameters the reference's albumentations pipeline draws (utilities/dataloaders.py:98-104) and launches the
one gather kernel that turns a decoded uint8 batch in HBM into the normalised fp32 NCHW batch + int64 masks the
training step consumes.  Decoding the .jpg / .png files stays on the host (cv2.imread in the reference, :76-78).

Probabilities and ranges (dataloaders.py:98-104 with albumentations' defaults where the reference passes none):
  HorizontalFlip p=0.4, VerticalFlip p=0.4, RandomBrightnessContrast p=0.5 with brightness, contrast ~ U(0, 0.5),
  Blur p=0.3 with an odd kernel size in {3, 5, 7}, RandomRotate90 p=0.3 with k ~ {0,1,2,3}, GaussNoise p=0.3 with
  variance ~ U(10, 50).
The draws come from a numpy Generator owned by the caller, not from albumentations' RNG streams (which are not
reproducible here): same distributions, different sequence.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # dataloaders.py:95
IMAGENET_STD = (0.229, 0.224, 0.225)       # dataloaders.py:96


class AugParams(NamedTuple):
    ints: np.ndarray       # [B, 8] int32: hflip, vflip, rot_k, blur_k, noise_on, 0, 0, 0
    floats: np.ndarray     # [B, 4] float32: alpha, beta, sigma, 0


def identity_params(B: int) -> AugParams:
    """t_val / t_test (dataloaders.py:106-107): resize only."""
    f = np.zeros((B, 4), np.float32)
    f[:, 0] = 1.0
    return AugParams(np.zeros((B, 8), np.int32), f)


def draw_train_params(B: int, rng: np.random.Generator) -> AugParams:
    """One draw of the t_train pipeline (dataloaders.py:98-104) per image."""
    i = np.zeros((B, 8), np.int32)
    f = np.zeros((B, 4), np.float32)
    f[:, 0] = 1.0
    for b in range(B):
        i[b, 0] = rng.random() < 0.4                                     # HorizontalFlip(p=0.4)
        i[b, 1] = rng.random() < 0.4                                     # VerticalFlip(p=0.4)
        if rng.random() < 0.5:                                           # RandomBrightnessContrast((0,0.5),(0,0.5)), p=0.5
            f[b, 0] = 1.0 + rng.uniform(0.0, 0.5)                        # alpha = 1 + contrast
            f[b, 1] = rng.uniform(0.0, 0.5)                              # beta (x max value 255: brightness_by_max)
        if rng.random() < 0.3:                                           # Blur(p=0.3), blur_limit 7 -> odd k in [3, 7]
            i[b, 3] = int(rng.choice([3, 5, 7]))
        if rng.random() < 0.3:                                           # RandomRotate90(p=0.3)
            i[b, 2] = int(rng.integers(0, 4))
        if rng.random() < 0.3:                                           # GaussNoise(p=0.3), var_limit (10, 50)
            i[b, 4] = 1
            f[b, 2] = float(np.sqrt(rng.uniform(10.0, 50.0)))
    return AugParams(i, f)


def augment_batch(images: torch.Tensor, masks: Optional[torch.Tensor], params: AugParams, out_size: Tuple[int, int] = (256, 256),
                  mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD, seed: int = 0,
                  noise: Optional[torch.Tensor] = None):
    """images: uint8 [B, Hs, Ws, 3] (RGB) on the GPU; masks: uint8 [B, Hs, Ws] on the GPU or None.
    Returns (x [B,3,Ho,Wo] float32 normalised, y [B,Ho,Wo] int64 or None).  `noise` ([B,3,Ho,Wo] float32, already
    scaled by each image's sigma) replaces the in-kernel Philox draw (tests use it to compare with the numpy oracle)."""
    _lib.require_device(images, "augment_batch")
    if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[3] != 3:
        raise ValueError("augment_batch: images must be uint8 [B, H, W, 3]")
    B, Hs, Ws, _ = images.shape
    Ho, Wo = out_size
    if (params.ints[:, 2] % 2 == 1).any() and Ho != Wo:
        raise ValueError("augment_batch: quarter turns need a square output")
    if masks is not None and (masks.dtype != torch.uint8 or tuple(masks.shape) != (B, Hs, Ws) or masks.device != images.device):
        raise ValueError("augment_batch: masks must be uint8 [B, H, W] on the images' device")
    if params.ints.shape != (B, 8) or params.floats.shape != (B, 4):
        raise ValueError("augment_batch: one parameter row per image expected")
    dev = images.device
    images = images.contiguous()
    masks = masks.contiguous() if masks is not None else None
    pi = torch.from_numpy(np.ascontiguousarray(params.ints, np.int32)).to(dev)
    pf = torch.from_numpy(np.ascontiguousarray(params.floats, np.float32)).to(dev)
    if noise is not None:
        if tuple(noise.shape) != (B, 3, Ho, Wo) or noise.dtype != torch.float32 or noise.device != dev:
            raise ValueError("augment_batch: noise must be float32 [B, 3, Ho, Wo] on the images' device")
        noise = noise.contiguous()
    x = torch.empty((B, 3, Ho, Wo), dtype=torch.float32, device=dev)
    y = torch.empty((B, Ho, Wo), dtype=torch.int64, device=dev) if masks is not None else None
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    with _lib.device_guard(dev):
        rc = _lib.lib().uaps_augment_batch(images.data_ptr(), masks.data_ptr() if masks is not None else None, pi.data_ptr(),
                                           pf.data_ptr(), noise.data_ptr() if noise is not None else None, int(seed), B, Hs, Ws,
                                           Ho, Wo, m3, s3, x.data_ptr(), y.data_ptr() if y is not None else None,
                                           _lib.current_stream(dev))
    _lib.check(rc, "uaps_augment_batch")
    return x, y
