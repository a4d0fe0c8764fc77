"""Every UAPS_* environment switch of the package in one place (round 6).

The C library reads no environment at all; the Python package reads each switch ONCE, through this module, when the module that owns
it is imported.  Almost all of them are A/B levers of the measurement scripts (tools/ab_*.sh, tools/ablation.sh) that default to the
shipped behaviour; `non_default()` names the ones a run changed, and bench.py prints that in its JSON line (`config.switches`) so
that a number can never silently come from a non-default configuration.

    flag(name, default)      "0" = off, anything else = on
    integer(name, default)
    text(name, default)
"""
from __future__ import annotations

import os
from typing import Dict

_seen: Dict[str, tuple] = {}      # name -> (default, value)

# what each switch is for (kept next to the one place that reads them)
DOC = {
    "UAPS_HIP_LIB": "path of another build of libuaps_hip.so with the same ABI (same-box A/B of two kernel builds)",
    "UAPS_CONV_MODE": "arithmetic of the convolutions: 2 / h16 (default), 1 / split (exact three-piece bf16), 0 / exact (fp32 MFMA)",
    "UAPS_DEFER_WRW_REDUCE": "weight-gradient reductions batched behind the backward (conv.deferred_reduces)",
    "UAPS_EARLY_WRW_REDUCE": "... and the decoders' share of them on a side stream beside the encoder's backward",
    "UAPS_EARLY_ADAM": "the decoders' Adam step behind their early reductions (world size 1)",
    "UAPS_FUSED_UP2": "bilinear x2 formed in the staging of up4's first convolution instead of materialised",
    "UAPS_FUSED_BN_SUMS": "BatchNorm-backward sums in the epilogue of the 16 -> 16 input-gradient row kernel (measured a wash: off)",
    "UAPS_LAZY_BN_BWD": "BatchNorm backward in two halves on the 256-wide decoder layers (lazybn)",
    "UAPS_PAIR_CFG": "cfg bits of the pair-loss kernels (diagnosis)",
    "UAPS_FUSED_FANOUT": "all perturbed copies of a feature map written by one pass over it",
    "UAPS_EPILOGUE_STATS": "BatchNorm statistics in the producing convolution's epilogue",
    "UAPS_VIRTUAL_CAT": "two-source convolutions instead of a materialised torch.cat in the UpBlocks",
    "UAPS_FUSED_FAN": "perturbation backward fused into the gradient fan-in kernel",
    "UAPS_DECODER_CHAINS": "decoders are dealt round-robin onto this many streams (diagnosis)",
    "UAPS_FAN_BESIDE": "perturbed feature copies written on a side stream beside the encoder's next levels",
    "UAPS_PACK_BESIDE": "the decoders' weights packed on a side stream beside the encoder's forward",
    "UAPS_FUSED_POOL": "2x2 max-pool fused into the fan-out / fan-in kernels",
    "UAPS_FUSED_BN_CONV": "BatchNorm + LeakyReLU applied in the consuming convolution's staging",
    "UAPS_DECODER_STREAMS": "one HIP stream per auxiliary decoder (bench.py switches it on for the headline)",
    "UAPS_STAT_SHIFT": "BatchNorm partial sums formed about running_mean - conv bias",
    "UAPS_INPUT_IN_GRAPH": "the captured step reads the batch from its own static input buffers (filled by the concatenation kernel)",
}


def _raw(name: str):
    return os.environ.get(name)


def flag(name: str, default: bool) -> bool:
    v = _raw(name)
    val = default if v is None or v == "" else (v != "0")
    _seen[name] = (default, val)
    return val


def integer(name: str, default: int) -> int:
    v = _raw(name)
    val = default if v is None or v == "" else int(v)
    _seen[name] = (default, val)
    return val


def text(name: str, default=None):
    v = _raw(name)
    val = default if v is None or v == "" else v
    _seen[name] = (default, val)
    return val


def non_default() -> Dict[str, object]:
    """{switch: value} for every switch read so far whose value is not its default, plus every other UAPS_* variable of the
    environment that the package knows and that is set (planner switches of _lib._TUNE_ENV, test hooks of bench.py)."""
    out = {k: v for k, (d, v) in sorted(_seen.items()) if v != d}
    for k in sorted(os.environ):
        if k.startswith("UAPS_") and k not in _seen:
            out[k] = os.environ[k]
    return out
