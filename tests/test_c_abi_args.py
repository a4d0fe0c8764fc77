"""C-ABI argument checking and the host-only planning entry points of libuaps_hip.so (include/uaps_hip.h), callable
without a GPU: every call below returns before any HIP API is touched (validation failures) or never touches one
(size / plan queries)."""
import ctypes as C

import pytest

from uaps_amd import _lib

OK, EINVAL, ERANGE = 0, -1, -2


@pytest.fixture(scope="module")
def L():
    return _lib.lib()


def test_error_codes_match_the_header(L):
    import re, os
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "uaps_hip.h")).read()
    vals = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(UAPS_E?\w+)\s+\(?(-?\d+)\)?", hdr)}
    assert vals["UAPS_OK"] == OK and vals["UAPS_EINVAL"] == EINVAL and vals["UAPS_ERANGE"] == ERANGE
    assert b"range" in L.uaps_error_string(ERANGE) or len(L.uaps_error_string(ERANGE)) > 0


def test_null_and_empty_arguments_are_refused(L):
    z = None
    assert L.uaps_conv_fwd(z, z, z, z, 1, 8, 8, 8, 8, 3, 0, z) == EINVAL
    assert L.uaps_conv_bwd_data(z, z, z, 1, 8, 8, 8, 8, 3, 0, z) == EINVAL
    assert L.uaps_conv_fwd_bn(z, z, 0.01, 1, z, z, z, z, 1, 8, 8, 8, 8, 3, 0, z) == EINVAL
    assert L.uaps_conv_bwd_weight_partial_bn(z, z, z, 0.01, 1, 0, 1, 8, 8, 8, 8, 3, 0, z, 0, z) == EINVAL
    assert L.uaps_bn_finalize_train(z, 4, z, z, z, z, z, z, 0.1, 1e-5, 2, 8, 8, 8, 1, z, z, z, z) == EINVAL
    assert L.uaps_softmax_klmap_bwd(z, z, z, 1, 4, 8, 8, z, z, z) == EINVAL
    assert L.uaps_augment_batch(z, z, z, z, z, 0, 1, 8, 8, 8, 8, (C.c_float * 3)(), (C.c_float * 3)(), z, z, z) == EINVAL
    assert L.uaps_up_cat_fwd(z, z, z, 1, 0, 4, 8, 8, z) == EINVAL
    assert L.uaps_sum_tensors(z, 2, z, 16, z) == EINVAL


def test_pack_sizes_and_workspace_queries(L):
    nf, nb = C.c_size_t(), C.c_size_t()
    assert L.uaps_conv_pack_floats(16, 3, 3, C.byref(nf), C.byref(nb)) == OK
    # exact layout: K padded to 4 (Cin <= 4) / 16, N padded to 16; then the three-piece bf16 layout: 3 pieces x taps x
    # channel groups (8 channels each, padded to a multiple of 4 groups) x N x 16 bytes; then a 32-float header (partial maxima of |w|, scale,
    # 1 / scale) and the two-piece fp16 layout of the same shape
    assert nf.value == 9 * 4 * 16 + 5 * 9 * 4 * 16 * 4 + 32 and nb.value == 9 * 16 * 16 + 5 * 9 * 4 * 16 * 4 + 32
    assert L.uaps_conv_pack_floats(64, 128, 1, C.byref(nf), C.byref(nb)) == OK
    assert nf.value == 128 * 64 + 5 * 16 * 64 * 4 + 32 and nb.value == 64 * 128 + 5 * 8 * 128 * 4 + 32
    assert L.uaps_conv_pack_floats(0, 3, 3, C.byref(nf), C.byref(nb)) == EINVAL
    assert L.uaps_conv_pack_floats(16, 3, 5, C.byref(nf), C.byref(nb)) == EINVAL          # only 1x1 and 3x3 exist
    n = C.c_size_t()
    assert L.uaps_conv_wrw_workspace_bytes(32, 16, 16, 256, 256, 3, 0, C.byref(n)) == OK
    assert n.value == 512 * (9 * 16 * 16 + 16) * 4                                       # 512 pixel splits of one 16x16 channel block
    assert L.uaps_conv_wrw_workspace_bytes(32, 16, 16, 256, 256, 5, 0, C.byref(n)) == EINVAL
    assert L.uaps_bn_workspace_bytes(0, 16, 8, 8, C.byref(n)) == EINVAL


def test_kernel_plan_names_and_statistics_tiles(L):
    buf = C.create_string_buffer(96)
    assert L.uaps_conv_set_mode(0) == OK and L.uaps_conv_get_mode() == 0
    assert L.uaps_conv_fwd_variant(32, 128, 128, 32, 32, 3, 0, buf, 96) == OK
    assert buf.value.decode() == "conv_fwd_kernel<3, 8, 32, 32, 8, 4, 1>"                # exact mode: the fp32 matrix instruction
    assert L.uaps_conv_set_mode(2) == OK and L.uaps_conv_get_mode() == 2                 # default: fp16 pieces where operands carry bounds
    assert L.uaps_conv_set_mode(1) == OK and L.uaps_conv_get_mode() == 1
    assert L.uaps_conv_fwd_variant(32, 128, 128, 32, 32, 3, 0, buf, 96) == OK
    assert buf.value.decode() == "conv_s32_kernel<32>"                                   # split modes: the 32x32x16 form (bf16 / fp16 pieces)
    assert L.uaps_conv_set_mode(2) == OK and L.uaps_conv_fwd_variant(32, 128, 128, 32, 32, 3, 0, buf, 96) == OK
    assert buf.value.decode() == "conv_sfwd_kernel<3, 8, 32, 32, 16>"                    # round 5: 128 channels on 32 x 32 maps with fp16 pieces: the 16x16x32 form
    assert L.uaps_conv_set_mode(1) == OK
    assert L.uaps_conv_fwd_variant(32, 128, 128, 32, 32, 3, 1 << 28, buf, 96) == OK
    assert buf.value.decode() == "conv_fwd_kernel<3, 8, 32, 32, 8, 4, 1>"                # cfg bit 28: exact kernels for this call
    assert L.uaps_conv_set_mode(7) == EINVAL
    assert L.uaps_conv_fwd_variant(32, 3, 16, 256, 256, 3, 0, buf, 96) == OK
    assert buf.value.decode() == "conv_fwd_kernel<3, 8, 32, 16, 4, 4, 1>"                # 3 input channels: 4-channel chunks
    assert L.uaps_conv_fwd_variant(2, 16, 16, 16, 16, 3, 0, buf, 96) == OK
    assert "16, 16" in buf.value.decode()                                                # narrow maps: 16x16 pixel tiles
    assert L.uaps_conv_fwd_variant(32, 64, 64, 64, 64, 3, 2 << 24, buf, 96) == OK
    assert buf.value.decode().endswith(", 2>")                                           # dilation 2 (ResNet stages)
    assert L.uaps_conv_fwd_variant(32, 64, 64, 64, 64, 3, 3 << 24, buf, 96) == ERANGE     # dilation 3 does not exist
    assert L.uaps_conv_wrw_variant(32, 64, 64, 64, 64, 3, 0, buf, 96) == OK
    assert buf.value.decode() == "conv_swrw_kernel<4, 2, 2>"                             # split mode: 32 x 32 channel blocks, 4-row tiles
    assert L.uaps_conv_wrw_variant(32, 64, 64, 64, 64, 3, 1 << 28, buf, 96) == OK
    assert buf.value.decode() == "conv_wrw_kernel<3, 4, 32, 2, 2, 4, 1>"                 # cfg bit 28: the fp32 matrix instruction
    assert L.uaps_conv_set_mode(2) == OK
    assert L.uaps_conv_fwd_variant(32, 16, 4, 256, 256, 3, 0, buf, 96) == OK and buf.value.decode() == "conv_small_kernel<8, 4>"      # class dimension
    assert L.uaps_conv_wrw_variant(32, 16, 4, 256, 256, 3, 0, buf, 96) == OK and buf.value.decode() == "conv_small_wrw_kernel"
    n = C.c_size_t()
    assert L.uaps_conv_wrw_workspace_bytes(32, 16, 4, 256, 256, 3, 0, C.byref(n)) == OK and n.value == 512 * (9 * 4 * 16 + 4) * 4
    parts = C.c_int()
    assert L.uaps_conv_fwd_stats_parts(32, 16, 16, 256, 256, 3, 0, C.byref(parts)) == OK and parts.value == 32 * 8
    assert L.uaps_conv_fwd_stats_parts(4, 128, 128, 16, 16, 3, 0, C.byref(parts)) == OK and parts.value == 1


def test_call_hints_are_validated_and_one_shot(L):
    h = _lib.CallHints()
    assert L.uaps_next_call_hints(C.byref(h)) == EINVAL                                  # struct_size 0: not a versioned struct
    h.struct_size = C.sizeof(_lib.CallHints) + 8
    assert L.uaps_next_call_hints(C.byref(h)) == EINVAL                                  # a client newer than the library
    h.struct_size = _lib.CallHints.out_amax.offset                                       # an old client's shorter struct: the tail reads as zero
    h.dyt_y = 1 << 20                                                                    # (garbage beyond its size is never looked at)
    assert L.uaps_next_call_hints(C.byref(h)) == OK
    h.dyt_y = None
    h.struct_size = C.sizeof(_lib.CallHints)
    assert L.uaps_next_call_hints(C.byref(h)) == OK
    h.bound[0], h.mul[0] = 1 << 20, 0.0                                                  # a bound needs a positive finite factor
    assert L.uaps_next_call_hints(C.byref(h)) == EINVAL
    h.mul[0] = float("inf")
    assert L.uaps_next_call_hints(C.byref(h)) == EINVAL
    h.mul[0] = 2.0
    assert L.uaps_next_call_hints(C.byref(h)) == OK
    assert L.uaps_conv_fwd(None, None, None, None, 1, 8, 8, 8, 8, 3, 0, None) == EINVAL   # consumes the pending hints
    assert L.uaps_next_call_hints(None) == OK
    # launch events: disarming with nothing armed reports "not consumed"; an armed pair that no launch used is dropped
    L.uaps_next_launch_events.restype = C.c_int
    L.uaps_next_launch_events.argtypes = [C.c_void_p, C.c_void_p]
    assert L.uaps_next_launch_events(None, None) == 0
    assert L.uaps_next_launch_events(C.c_void_p(16), C.c_void_p(32)) == 0
    assert L.uaps_next_launch_events(None, None) == 0
    out = (C.c_float * 1024)()
    assert L.uaps_bn_param_bounds(None, None, None, 1, out, None) == EINVAL


def test_tensors_of_two_gib_or_more_are_refused_before_any_launch(L):
    """The conv kernels address a channel block with 32-bit buffer offsets: a [C, H, W] block of >= 2 GiB is out of range.
    The check precedes every HIP call, so dummy (never dereferenced) pointers suffice here."""
    fake = C.c_void_p(1 << 20)
    big = (1, 16, 16, 8192, 8192)                       # 16 * 8192 * 8192 * 4 B = 4 GiB per image
    assert L.uaps_conv_fwd(fake, fake, None, fake, *big, 3, 0, None) == ERANGE
    assert L.uaps_conv_bwd_data(fake, fake, fake, *big, 3, 0, None) == ERANGE
    assert L.uaps_conv_bwd_weight_partial(fake, fake, 0, *big, 3, 0, fake, 1 << 40, None) == ERANGE
    assert L.uaps_conv_fwd(fake, fake, None, fake, 1, 16, 16, 64, 64, 5, 0, None) == ERANGE      # 5x5 does not exist


def test_process_wide_switches_and_the_round3_plans(L):
    """uaps_conv_set_tuning / uaps_set_error_word (the library reads no environment itself), and the plans added for the
    ResNet-50 configuration: GEMM-tiled 1x1 kernels, dilated split kernels, the stem weight gradient's workspace, the
    sampling-phase helpers."""
    import re, os
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "uaps_hip.h")).read()
    tune = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(UAPS_TUNE_\w+)\s+\(?(\d+)u?\)?", hdr)}
    assert tune["UAPS_TUNE_NO_SPLIT_FWD"] == 1 and tune["UAPS_TUNE_NO_SPLIT_WRW"] == 2 and tune["UAPS_TUNE_DEEP_ROWS"] == 512 and tune["UAPS_TUNE_G1_NARROW"] == 1024 and tune["UAPS_TUNE_NO_G"] == 2048 and tune["UAPS_TUNE_G_DEEP"] == 4096 and len(tune) == 13
    L.uaps_conv_get_tuning.restype = C.c_uint
    buf, parts, n = C.create_string_buffer(96), C.c_int(), C.c_size_t()
    prev_mode, prev_tune = L.uaps_conv_get_mode(), L.uaps_conv_get_tuning()
    try:
        assert L.uaps_conv_set_mode(1) == OK and L.uaps_conv_set_tuning(0) == OK
        wide = (16, 256, 64, 160, 160, 1, 0)                                             # layer1's 1x1 reduction at configs[4]'s shape
        assert L.uaps_conv_fwd_variant(*wide, buf, 96) == OK and buf.value.decode() == "conv_g1s_kernel<64>"
        assert L.uaps_conv_fwd_stats_parts(*wide, C.byref(parts)) == OK and parts.value == 160 * 160 // 128
        assert L.uaps_conv_wrw_variant(*wide, buf, 96) == OK and buf.value.decode() == "conv_gw1s_kernel"
        assert L.uaps_conv_fwd_variant(16, 512, 512, 80, 80, 3, 4 << 24, buf, 96) == OK and buf.value.decode() == "conv_s32d_kernel<64, 4>"
        assert L.uaps_conv_set_tuning(tune["UAPS_TUNE_NO_SPLIT_FWD"] | tune["UAPS_TUNE_NO_SPLIT_WRW"]) == OK
        assert L.uaps_conv_get_tuning() == 3
        assert L.uaps_conv_fwd_variant(*wide, buf, 96) == OK and buf.value.decode().startswith("conv_fwd_kernel<1,")
        assert L.uaps_conv_fwd_stats_parts(*wide, C.byref(parts)) == OK and parts.value == 20 * 5      # 8 x 32 pixel tiles again
        assert L.uaps_conv_wrw_variant(*wide, buf, 96) == OK and buf.value.decode().startswith("conv_wrw_kernel<1,")
        assert L.uaps_conv_fwd_variant(16, 512, 512, 80, 80, 3, 4 << 24, buf, 96) == OK and buf.value.decode().endswith(", 4>") \
            and buf.value.decode().startswith("conv_fwd_kernel<3,")
    finally:
        L.uaps_conv_set_mode(prev_mode)
        L.uaps_conv_set_tuning(prev_tune)
    assert L.uaps_set_error_word(C.c_void_p(2)) == EINVAL                                # a 4-byte word
    assert L.uaps_set_error_word(None) == OK                                             # detaches (nothing is reported)
    # the stem weight gradient (7x7 / 2 / 3, <= 3 input channels, OW % 4 == 0): 1024 splits of [64 channels][160 columns]
    assert L.uaps_convs_wrw_workspace_bytes(16, 3, 64, 640, 640, 7, 2, 3, C.byref(n)) == OK and n.value == 1024 * 64 * 160 * 4
    assert L.uaps_convs_wrw_workspace_bytes(2, 3, 64, 50, 70, 7, 2, 3, C.byref(n)) == OK and n.value % (49 * 64 * 16 * 4) == 0      # odd width: the general kernel's slabs
    fake = C.c_void_p(1 << 20)
    assert L.uaps_space_to_depth2(None, fake, 1, 8, 8, 8, 0, None) == EINVAL
    assert L.uaps_space_to_depth2(fake, fake, 1, 8, 7, 8, 0, None) == ERANGE             # odd height
    assert L.uaps_space_to_depth2(fake, fake, 1, 8, 8, 12, 1, None) == ERANGE            # rows of 8-float groups
    assert L.uaps_subsample2_fwd(None, fake, 4, 8, 8, None) == EINVAL and L.uaps_subsample2_bwd(fake, None, 4, 8, 8, None) == EINVAL
    assert L.uaps_subsample2_fwd(fake, fake, 0, 8, 8, None) == EINVAL


def test_explicit_hints_forms_validate_their_record(L):
    """The *_h entry points (include/uaps_hip.h, "Explicit-hints forms"): the hints record is an argument, validated like
    uaps_next_call_hints validates it, and a pending thread-local record is not consumed by them."""
    assert set(_lib.HINTED) <= set(_lib.SIGNATURES) and all(n + "_h" in _lib.SIGNATURES for n in _lib.HINTED)
    h = _lib.CallHints()
    h.struct_size = C.sizeof(_lib.CallHints) + 8                                         # a client newer than the library
    assert L.uaps_conv_fwd_h(C.byref(h), None, None, None, None, 1, 8, 8, 8, 8, 3, 0, None) == EINVAL
    h.struct_size = C.sizeof(_lib.CallHints)
    h.bound[0], h.mul[0] = 1 << 20, -1.0                                                 # a bound needs a positive finite factor
    assert L.uaps_conv_bwd_data_h(C.byref(h), None, None, None, 1, 8, 8, 8, 8, 3, 0, None) == EINVAL
    h.mul[0] = 1.0
    fake = C.c_void_p(1 << 20)
    big = (1, 16, 16, 8192, 8192)
    assert L.uaps_conv_fwd_h(C.byref(h), fake, fake, None, fake, *big, 3, 0, None) == ERANGE      # argument checks as in the legacy form
    assert L.uaps_conv_fwd_h(None, None, None, None, None, 1, 8, 8, 8, 8, 3, 0, None) == EINVAL   # NULL hints = none
    h.struct_size = 0                                                                    # struct_size 0 = no hints either
    assert L.uaps_cat2_h(C.byref(h), None, None, None, 4, None) == EINVAL                # (then the NULL tensors are refused)
    # a pending legacy record survives explicit calls and is consumed by the next legacy call only
    p = _lib.CallHints()
    p.struct_size = C.sizeof(_lib.CallHints)
    assert L.uaps_next_call_hints(C.byref(p)) == OK
    assert L.uaps_conv_fwd_h(None, None, None, None, None, 1, 8, 8, 8, 8, 3, 0, None) == EINVAL
    assert L.uaps_conv_fwd(None, None, None, None, 1, 8, 8, 8, 8, 3, 0, None) == EINVAL
    assert L.uaps_next_call_hints(None) == OK
    # uaps_conv_call: `stream` sits in front of the growable hints record (ABI 3)
    assert _lib.ConvCall.stream.offset < _lib.ConvCall.hints.offset
    assert _lib.ConvCall.hints.offset + C.sizeof(_lib.CallHints) == C.sizeof(_lib.ConvCall)
