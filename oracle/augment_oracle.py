"""CPU oracle of the input pipeline.  TEST INFRASTRUCTURE, NOT PRODUCT.

PARITY UNPINNED: the reference's loader (utilities/dataloaders.py:60-119) runs cv2 + albumentations, neither of which is
in the build image, and the reference ships no fixtures of augmented batches.  This file restates, in plain numpy, the
documented uint8 behaviour of the calls the loader makes, stage by stage and in the loader's order:

  :98  A.Resize(256, 256, interpolation=cv2.INTER_NEAREST)   src index = floor(dst * src_size / dst_size)
  :98  A.HorizontalFlip / A.VerticalFlip                       [:, ::-1] / [::-1]
  :99  A.RandomBrightnessContrast                              uint8 LUT: clip(v * alpha + beta * 255, 0, 255).astype(uint8)
  :100 A.Blur                                                  cv2.blur(k x k), BORDER_REFLECT_101, uint8 result rounded half-to-even
  :101 A.RandomRotate90                                        np.rot90(img, k)
  :102 A.GaussNoise                                            float32(img) + gauss, clip to [0, 255], astype(uint8)
  :90  T.ToTensor + T.Normalize(mean, std)                     v / 255, (v - mean) / std, HWC -> CHW
  :92  mask -> torch.long                                      geometric stages only (nearest)
Only tests/ may import it.
"""
import numpy as np


def resize_nearest(img, Ho, Wo):
    Hs, Ws = img.shape[:2]
    ys = np.minimum((np.arange(Ho) * Hs) // Ho, Hs - 1)
    xs = np.minimum((np.arange(Wo) * Ws) // Wo, Ws - 1)
    return img[ys][:, xs]


def brightness_contrast(img, alpha, beta):
    lut = np.arange(256, dtype=np.float32) * np.float32(alpha) + np.float32(beta) * np.float32(255.0)
    lut = np.clip(lut, 0, 255).astype(np.uint8)
    return lut[img]


def box_blur(img, k):
    r = k // 2
    p = np.pad(img.astype(np.float32), ((r, r), (r, r), (0, 0)), mode="reflect")     # numpy 'reflect' == BORDER_REFLECT_101
    H, W = img.shape[:2]
    acc = np.zeros(img.shape, np.float32)
    for dy in range(k):
        for dx in range(k):
            acc += p[dy:dy + H, dx:dx + W]
    return np.rint(acc * np.float32(1.0 / (k * k))).astype(np.uint8)                 # rint: half to even, like cvRound


def augment_one(img, mask, ints, floats, noise, Ho, Wo, mean, std):
    """img uint8 [Hs,Ws,3], mask uint8 [Hs,Ws] or None, ints/floats = one row of AugParams, noise float32 [3,Ho,Wo]."""
    hflip, vflip, rot, bk, noise_on = (int(v) for v in ints[:5])
    alpha, beta = float(floats[0]), float(floats[1])
    x = resize_nearest(img, Ho, Wo)
    m = resize_nearest(mask, Ho, Wo) if mask is not None else None
    if hflip:
        x = x[:, ::-1]; m = m[:, ::-1] if m is not None else None
    if vflip:
        x = x[::-1]; m = m[::-1] if m is not None else None
    x = brightness_contrast(x, alpha, beta)
    if bk:
        x = box_blur(x, bk)
    x = np.rot90(x, rot); m = np.rot90(m, rot) if m is not None else None
    if noise_on:
        x = np.clip(x.astype(np.float32) + np.transpose(noise, (1, 2, 0)), 0, 255).astype(np.uint8)
    t = x.astype(np.float32) / np.float32(255.0)
    t = (t - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(np.transpose(t, (2, 0, 1))), (np.ascontiguousarray(m).astype(np.int64) if m is not None else None)
