"""Deterministic, RNG-free parameter values for large models (shared by tools/make_golden.py, which fills the
reference's classes with them, and by the tests, which fill this build's classes): a fixture then only has to hold
the input and the outputs, not a 100 MB state_dict."""
import zlib

import numpy as np
import torch


def formula_state_dict(sd):
    """New values for every entry of a state_dict, derived from the entry's name and shape only."""
    out = {}
    for name, t in sd.items():
        n = t.numel()
        ph = (zlib.crc32(name.encode()) % 1000) / 1000.0 * 6.283
        idx = np.arange(n, dtype=np.float64)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.tensor(3, dtype=t.dtype)
            continue
        if name.endswith("running_var"):
            v = 1.0 + 0.3 * np.abs(np.sin(0.37 * idx + ph))
        elif name.endswith("running_mean"):
            v = 0.05 * np.sin(0.53 * idx + ph)
        elif t.dim() == 1 and name.endswith("weight"):          # BatchNorm gamma
            v = 1.0 + 0.1 * np.sin(0.41 * idx + ph)
        elif t.dim() == 1:                                       # biases / beta
            v = 0.05 * np.cos(0.29 * idx + ph)
        else:                                                    # conv weights: He-like scale
            fan_in = int(np.prod(t.shape[1:]))
            v = np.sin(0.7389 * idx + ph) * np.sqrt(2.0 / fan_in)
        out[name] = torch.tensor(v.reshape(tuple(t.shape)), dtype=t.dtype)
    return out


def formula_input(shape, k=0.0173):
    n = int(np.prod(shape))
    return torch.tensor((np.sin(k * np.arange(n, dtype=np.float64) ** 1.1) * 1.5).reshape(shape), dtype=torch.float32)
