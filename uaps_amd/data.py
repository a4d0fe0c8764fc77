"""Synthetic on-device batch source with the shapes/dtypes of the reference's loaders
(utilities/dataloaders.py:60-119: ImageNet-normalised fp32 images [B,3,H,W], int64 masks [B,H,W]).
The dataset itself is not shipped with the reference and the input pipeline is out of scope
(SURVEY.md section 8d), so benchmarks and smoke tests use this generator."""
from __future__ import annotations

import numpy as np
import torch


def synthetic_masks(rng: np.random.Generator, B: int, C: int, H: int, W: int) -> np.ndarray:
    """Background 0 with 1-3 axis-aligned rectangles of classes 1..C-1 covering roughly 10-20 % of the
    image (NEU-like sparsity)."""
    y = np.zeros((B, H, W), np.int64)
    for b in range(B):
        for _ in range(int(rng.integers(1, 4))):
            c = int(rng.integers(1, C)) if C > 1 else 0
            rh, rw = int(rng.integers(H // 8, H // 3 + 1)), int(rng.integers(W // 8, W // 3 + 1))
            h0, w0 = int(rng.integers(0, H - rh + 1)), int(rng.integers(0, W - rw + 1))
            y[b, h0:h0 + rh, w0:w0 + rw] = c
    return y


class SyntheticBatches:
    """`n_batches` labelled + unlabelled batch pairs generated once, kept resident in HBM and cycled."""

    def __init__(self, batch: int, in_chns: int = 3, num_classes: int = 4, H: int = 256, W: int = 256,
                 n_batches: int = 2, seed: int = 1337, device="cuda"):
        rng = np.random.default_rng(seed)
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.items = []
        for _ in range(n_batches):
            xl = torch.randn((batch, in_chns, H, W), generator=g, dtype=torch.float32)
            xu = torch.randn((batch, in_chns, H, W), generator=g, dtype=torch.float32)
            yl = torch.from_numpy(synthetic_masks(rng, batch, num_classes, H, W))
            self.items.append((xl.to(device), yl.to(device), xu.to(device)))
        self.i = 0

    def next(self):
        it = self.items[self.i % len(self.items)]
        self.i += 1
        return it
