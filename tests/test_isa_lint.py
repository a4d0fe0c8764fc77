"""The shipped code objects contain no packed-fp32 instruction that selects the low half of its second source from the high
register (`op_sel:[_,1,..]`): on MI355X that form returned wrong low halves for lanes 48..63 whenever another wave of the SIMD
issued v_mfma_f32_16x16x32 (tools/diag/pkfma_probe.hip) -- the root cause of the run-to-run differences of
the training step with decoder streams or a second process on the card."""
import os
import sys

import pytest

from conftest import ROOT


def test_no_packed_fp32_low_half_swizzle_on_src1():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    so = os.path.join(ROOT, "uaps_amd", "lib", "libuaps_hip.so")
    if not os.path.exists(so):
        pytest.fail("libuaps_hip.so not built")
    if not os.path.exists(os.path.join(isa_lint.LLVM, "llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    flagged, info, n = isa_lint.lint(so)
    assert n > 1000, "the disassembly found no packed fp32 instructions: wrong file?"
    assert not flagged, f"{len(flagged)} forbidden instructions, e.g. {flagged[:3]} (python tools/isa_lint.py lists them per kernel)"
    # low-half swizzles of the first / third source were clean in the probe (2e9 lane-iterations each), but nothing needs them
    # either: none is shipped, which keeps the question closed
    assert not info, f"{len(info)} packed fp32 instructions with a low-half op_sel on src0 / src2, e.g. {info[:3]}"
