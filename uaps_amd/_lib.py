"""ctypes binding of libuaps_hip.so (C ABI: include/uaps_hip.h).

There is no fallback: if the shared library is missing or a symbol is absent, every op that needs it
raises.  `python -c "import __graft_entry__ as g; g.build()"` (or `make -C uaps_amd/csrc`) builds it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

from . import config

_HERE = os.path.dirname(os.path.abspath(__file__))
# UAPS_HIP_LIB selects another build of the same ABI (tools/ab_bench.sh compares two kernel builds on one GPU box)
LIB_PATH = config.text("UAPS_HIP_LIB") or os.path.join(_HERE, "lib", "libuaps_hip.so")
CSRC = os.path.join(_HERE, "csrc")

_lock = threading.Lock()
_tls = threading.local()      # one reusable uaps_call_hints ARGUMENT buffer per Python thread (the *_h entry points copy it during the call)
_lib = None

c_float_p = C.POINTER(C.c_float)
c_void_p = C.c_void_p

# symbol -> (restype, argtypes); mirrors include/uaps_hip.h one to one
_PTR = C.c_void_p
SIGNATURES = {
    "uaps_abi_version": (C.c_int, []),
    "uaps_error_string": (C.c_char_p, [C.c_int]),
    "uaps_loss_workspace_bytes": (C.c_int, [C.c_int] * 5 + [C.POINTER(C.c_size_t)]),
    "uaps_unsup_fwd": (C.c_int, [_PTR, _PTR] + [C.c_int] * 5 + [C.c_float] * 3 + [_PTR, _PTR, _PTR, _PTR, C.c_size_t, _PTR]),
    "uaps_unsup_bwd": (C.c_int, [_PTR, _PTR, _PTR, C.c_float, C.c_float, _PTR] + [C.c_int] * 5 + [_PTR, _PTR]),
    "uaps_sup_fwd": (C.c_int, [_PTR, _PTR] + [C.c_int] * 5 + [C.c_float] * 3 + [_PTR, _PTR, C.c_size_t, _PTR]),
    "uaps_sup_bwd": (C.c_int, [_PTR, _PTR, _PTR, C.c_float, C.c_float, _PTR] + [C.c_int] * 5 + [_PTR, _PTR]),
    "uaps_pairloss_workspace_bytes": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "uaps_pairloss_num_sums": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "uaps_pairloss_fwd": (C.c_int, [_PTR] * 4 + [C.c_int] * 5 + [C.c_float] * 3 + [_PTR] * 6 + [C.c_size_t, C.c_int, _PTR]),
    "uaps_pairloss_finalize_sums": (C.c_int, [_PTR, C.c_int, C.c_int, C.c_long] + [C.c_float] * 3 + [_PTR] * 3),
    "uaps_pairloss_bwd": (C.c_int, [_PTR] * 6 + [C.c_float] * 2 + [_PTR] + [C.c_int] * 5 + [C.c_long, _PTR, _PTR, C.c_int, _PTR]),
    "uaps_feat_noise": (C.c_int, [_PTR, _PTR] + [C.c_int] * 4 + [C.c_uint64, C.c_uint64, C.c_float, _PTR, _PTR]),
    "uaps_feat_noise_apply": (C.c_int, [_PTR, _PTR, _PTR, C.c_int, C.c_long, _PTR]),
    "uaps_feat_bernoulli": (C.c_int, [_PTR, _PTR, C.c_long, C.c_uint64, C.c_uint64, C.c_float, _PTR, _PTR]),
    "uaps_feat_mask_apply": (C.c_int, [_PTR, _PTR, C.c_float, _PTR, C.c_long, _PTR]),
    "uaps_feat_dropout_workspace_bytes": (C.c_int, [C.c_int] * 4 + [C.POINTER(C.c_size_t)]),
    "uaps_feat_dropout_fwd": (C.c_int, [_PTR, _PTR] + [C.c_int] * 4 + [C.c_float, _PTR, _PTR, C.c_size_t, _PTR]),
    "uaps_feat_dropout_bwd": (C.c_int, [_PTR, _PTR, _PTR] + [C.c_int] * 4 + [_PTR]),
    "uaps_bn_workspace_bytes": (C.c_int, [C.c_int] * 4 + [C.POINTER(C.c_size_t)]),
    "uaps_bn_act_fwd_train": (C.c_int, [_PTR] * 7 + [C.c_float] * 4 + [C.c_uint64, C.c_uint64] + [C.c_int] * 4 + [_PTR, _PTR, _PTR, _PTR, C.c_size_t, _PTR]),
    "uaps_bn_act_fwd_train_grouped": (C.c_int, [_PTR] * 7 + [C.c_float] * 4 + [C.c_uint64, C.c_uint64] + [C.c_int] * 5 + [_PTR, _PTR, _PTR, _PTR, C.c_size_t, _PTR]),
    "uaps_bn_act_fwd_train_partials": (C.c_int, [_PTR, C.c_int] + [_PTR] * 7 + [C.c_float] * 4 + [C.c_uint64, C.c_uint64] + [C.c_int] * 5 + [_PTR, _PTR, _PTR, _PTR, C.c_size_t, _PTR]),
    "uaps_bn_act_bwd_grouped": (C.c_int, [_PTR] * 6 + [C.c_float] * 2 + [C.c_uint64, C.c_uint64] + [C.c_int] * 5 + [_PTR, _PTR, _PTR, _PTR, C.c_size_t, _PTR]),
    "uaps_bn_act_bwd_grouped_bias": (C.c_int, [_PTR] * 6 + [C.c_float] * 2 + [C.c_uint64, C.c_uint64] + [C.c_int] * 5 + [_PTR] * 5 + [C.c_size_t, _PTR]),
    "uaps_bn_act_fwd_eval": (C.c_int, [_PTR] * 6 + [C.c_float] * 2 + [C.c_int] * 4 + [_PTR, _PTR, _PTR, C.c_size_t, _PTR]),
    "uaps_bn_act_bwd": (C.c_int, [_PTR] * 6 + [C.c_float] * 2 + [C.c_uint64, C.c_uint64] + [C.c_int] * 4 + [_PTR, _PTR, _PTR, _PTR, C.c_size_t, _PTR]),
    "uaps_bn_act_bwd_eval": (C.c_int, [_PTR] * 6 + [C.c_float] * 2 + [C.c_int] * 4 + [_PTR, _PTR, C.c_size_t, _PTR]),
    "uaps_up_cat_fwd": (C.c_int, [_PTR] * 3 + [C.c_int] * 5 + [_PTR]),
    "uaps_up_cat_bwd": (C.c_int, [_PTR] * 3 + [C.c_int] * 5 + [_PTR]),
    "uaps_conv_set_mode": (C.c_int, [C.c_int]),
    "uaps_conv_get_mode": (C.c_int, []),
    "uaps_conv_set_tuning": (C.c_int, [C.c_uint]),
    "uaps_conv_get_tuning": (C.c_uint, []),
    "uaps_set_error_word": (C.c_int, [_PTR]),
    "uaps_conv_pack_floats": (C.c_int, [C.c_int] * 3 + [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "uaps_conv_pack_weights": (C.c_int, [_PTR] + [C.c_int] * 3 + [_PTR, _PTR, _PTR]),
    "uaps_conv_pack_weights_batch": (C.c_int, [_PTR] * 6 + [C.c_int, _PTR]),
    "uaps_conv_fwd": (C.c_int, [_PTR] * 4 + [C.c_int] * 7 + [_PTR]),
    "uaps_conv_fwd_stats": (C.c_int, [_PTR] * 5 + [C.c_int] * 7 + [_PTR]),
    "uaps_conv_fwd_stats_parts": (C.c_int, [C.c_int] * 7 + [C.POINTER(C.c_int)]),
    "uaps_conv_bwd_data": (C.c_int, [_PTR] * 3 + [C.c_int] * 7 + [_PTR]),
    "uaps_conv_wrw_workspace_bytes": (C.c_int, [C.c_int] * 7 + [C.POINTER(C.c_size_t)]),
    "uaps_conv_bwd_weight_partial": (C.c_int, [_PTR, _PTR] + [C.c_int] * 8 + [_PTR, C.c_size_t, _PTR]),
    "uaps_conv_bwd_weight_reduce": (C.c_int, [_PTR, _PTR, _PTR] + [C.c_int] * 7 + [_PTR]),
    "uaps_conv_bwd_weight_reduce_batch": (C.c_int, [_PTR, C.c_int, _PTR]),
    "uaps_conv_fwd_cat": (C.c_int, [_PTR, C.c_int, _PTR, C.c_int] + [_PTR] * 4 + [C.c_int] * 6 + [_PTR]),
    "uaps_conv_bwd_data_cat": (C.c_int, [_PTR, _PTR, _PTR, C.c_int, _PTR, C.c_int] + [C.c_int] * 6 + [_PTR]),
    "uaps_conv_bwd_weight_partial_cat": (C.c_int, [_PTR, _PTR, C.c_int, _PTR, C.c_int] + [C.c_int] * 7 + [_PTR, C.c_size_t, _PTR]),
    "uaps_bn_finalize_train": (C.c_int, [_PTR, C.c_int] + [_PTR] * 6 + [C.c_float] * 2 + [C.c_int] * 5 + [_PTR] * 4),
    "uaps_conv_fwd_bn": (C.c_int, [_PTR, _PTR, C.c_float, C.c_int] + [_PTR] * 4 + [C.c_int] * 7 + [_PTR]),
    "uaps_conv_bwd_weight_partial_bn": (C.c_int, [_PTR, _PTR, _PTR, C.c_float, C.c_int] + [C.c_int] * 8 + [_PTR, C.c_size_t, _PTR]),
    "uaps_conv_fwd_variant": (C.c_int, [C.c_int] * 7 + [C.c_char_p, C.c_size_t]),
    "uaps_conv_wrw_variant": (C.c_int, [C.c_int] * 7 + [C.c_char_p, C.c_size_t]),
    "uaps_conv_bwd_weight": (C.c_int, [_PTR] * 4 + [C.c_int] * 7 + [_PTR, C.c_size_t, _PTR]),
    "uaps_set_step_state": (C.c_int, [_PTR]),
    "uaps_next_call_hints": (C.c_int, [_PTR]),
    "uaps_conv_ex": (C.c_int, [_PTR]),
    "uaps_account": (C.c_int, [C.c_int]),
    "uaps_accounted_bytes": (C.c_double, []),
    "uaps_next_launch_events": (C.c_int, [_PTR, _PTR]),
    "uaps_zero_bounds": (C.c_int, [_PTR, C.c_long, _PTR]),
    "uaps_bn_param_bounds": (C.c_int, [_PTR, _PTR, _PTR, C.c_int, _PTR, _PTR]),
    "uaps_get_step_state": (C.c_void_p, []),
    "uaps_convs_pack_floats": (C.c_int, [C.c_int] * 3 + [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "uaps_convs_pack_weights": (C.c_int, [_PTR] + [C.c_int] * 3 + [_PTR, _PTR, _PTR]),
    "uaps_convs_out_size": (C.c_int, [C.c_int] * 5 + [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "uaps_convs_fwd": (C.c_int, [_PTR] * 3 + [C.c_int] * 8 + [_PTR]),
    "uaps_convs_bwd_data": (C.c_int, [_PTR] * 3 + [C.c_int] * 8 + [_PTR]),
    "uaps_convs_wrw_workspace_bytes": (C.c_int, [C.c_int] * 8 + [C.POINTER(C.c_size_t)]),
    "uaps_convs_bwd_weight": (C.c_int, [_PTR] * 3 + [C.c_int] * 8 + [_PTR, C.c_size_t, _PTR]),
    "uaps_maxpool3x3s2_fwd": (C.c_int, [_PTR, _PTR, _PTR, C.c_long, C.c_int, C.c_int, _PTR]),
    "uaps_maxpool3x3s2_bwd": (C.c_int, [_PTR, _PTR, _PTR, C.c_long, C.c_int, C.c_int, _PTR]),
    "uaps_space_to_depth2": (C.c_int, [_PTR, _PTR, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _PTR]),
    "uaps_subsample2_fwd": (C.c_int, [_PTR, _PTR, C.c_long, C.c_int, C.c_int, _PTR]),
    "uaps_subsample2_bwd": (C.c_int, [_PTR, _PTR, C.c_long, C.c_int, C.c_int, _PTR]),
    "uaps_sum_tensors": (C.c_int, [_PTR, C.c_int, _PTR, C.c_long, _PTR]),
    "uaps_pair_workspace_bytes": (C.c_int, [C.POINTER(C.c_size_t)]),
    "uaps_feat_dropout_stats": (C.c_int, [_PTR] + [C.c_int] * 4 + [_PTR, C.c_size_t, _PTR]),
    "uaps_fanout_perturbed": (C.c_int, [_PTR] * 7 + [C.c_int, C.c_int, C.c_uint64, C.c_float, C.c_float] + [C.c_int] * 4 + [_PTR]),
    "uaps_augment_batch": (C.c_int, [_PTR] * 5 + [C.c_uint64] + [C.c_int] * 5 + [C.POINTER(C.c_float), C.POINTER(C.c_float), _PTR, _PTR, _PTR]),
    "uaps_softmax_klmap_bwd": (C.c_int, [_PTR] * 3 + [C.c_int] * 4 + [_PTR] * 3),
    "uaps_softmax_pair_fwd": (C.c_int, [_PTR, _PTR] + [C.c_int] * 5 + [_PTR, _PTR, _PTR, _PTR, C.c_size_t, _PTR]),
    "uaps_softmax_mse_bwd": (C.c_int, [_PTR] * 3 + [C.c_int] * 4 + [_PTR, _PTR]),
    "uaps_softmax_kl_bwd": (C.c_int, [_PTR] * 3 + [C.c_int] * 4 + [_PTR, _PTR]),
    "uaps_entropy_map": (C.c_int, [_PTR] + [C.c_int] * 4 + [_PTR, _PTR, _PTR, C.c_size_t, _PTR]),
    "uaps_fanin_perturbed": (C.c_int, [_PTR] * 4 + [C.c_int, C.c_int, C.c_uint64, C.c_float, C.c_float] + [C.c_int] * 4 + [_PTR, _PTR]),
    "uaps_maxpool2x2_fwd": (C.c_int, [_PTR] + [C.c_int] * 4 + [_PTR, _PTR, _PTR]),
    "uaps_adam_step": (C.c_int, [_PTR] * 5 + [C.c_int] + [C.c_double] * 5 + [C.c_long, _PTR]),
    "uaps_add_relu": (C.c_int, [_PTR, _PTR, _PTR, C.c_long, _PTR]),
    "uaps_relu_bwd": (C.c_int, [_PTR, _PTR, _PTR, C.c_long, _PTR]),
    "uaps_relu_bwd_sum": (C.c_int, [_PTR, C.c_int, _PTR, _PTR, C.c_long, _PTR]),
    "uaps_cat2": (C.c_int, [_PTR, _PTR, _PTR, C.c_long, _PTR]),
    "uaps_bn_act_bwd_prepare": (C.c_int, [_PTR] * 6 + [C.c_float] + [C.c_int] * 5 + [_PTR] * 6 + [C.c_size_t, _PTR]),
    "uaps_bn_act_bwd_apply": (C.c_int, [_PTR] * 3 + [C.c_float] + [C.c_int] * 5 + [_PTR, _PTR]),
    "uaps_bn_act_bwd_finalize": (C.c_int, [_PTR, C.c_int] + [_PTR] * 5 + [C.c_int] * 5 + [_PTR] * 6),
    "uaps_seg_confusion": (C.c_int, [_PTR, _PTR] + [C.c_int] * 4 + [_PTR, _PTR]),
}


# The explicit-hints forms (include/uaps_hip.h, "Explicit-hints forms"): `<name>_h(const uaps_call_hints*, same arguments)`.  These are
# what the package calls; the base names stay bound for the C-ABI tests and foreign callers.
HINTED = ("uaps_conv_fwd", "uaps_conv_fwd_stats", "uaps_conv_fwd_bn", "uaps_conv_fwd_cat", "uaps_conv_bwd_data", "uaps_conv_bwd_data_cat",
          "uaps_conv_bwd_weight_partial", "uaps_conv_bwd_weight_partial_bn", "uaps_conv_bwd_weight_partial_cat",
          "uaps_bn_act_fwd_train_partials", "uaps_bn_finalize_train", "uaps_bn_act_bwd_grouped", "uaps_bn_act_bwd_grouped_bias",
          "uaps_bn_act_bwd_apply", "uaps_up_cat_fwd", "uaps_cat2", "uaps_add_relu", "uaps_pairloss_bwd")
for _n in HINTED:
    SIGNATURES[_n + "_h"] = (SIGNATURES[_n][0], [_PTR] + list(SIGNATURES[_n][1]))
del _n


class UapsHipError(RuntimeError):
    pass


def build(verbose: bool = False, jobs: int = 8) -> str:
    """Compile every HIP source under uaps_amd/csrc for gfx950 into uaps_amd/lib/libuaps_hip.so."""
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise UapsHipError("building libuaps_hip.so failed (see output above)")
    return LIB_PATH


def lib() -> C.CDLL:
    """The loaded library with typed entry points; raises UapsHipError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise UapsHipError(
                    f"{LIB_PATH} not found: the UAPS HIP kernels are not built. Run "
                    "`make -C uaps_amd/csrc` (needs hipcc; cross-compiles for gfx950 without a GPU). "
                    "uaps_amd has no CPU or PyTorch fallback for these ops.")
            l = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                try:
                    fn = getattr(l, name)
                except AttributeError as e:
                    raise UapsHipError(f"libuaps_hip.so does not export {name}; rebuild it") from e
                fn.restype, fn.argtypes = res, args
            _configure_from_environment(l)
            _lib = l
    return _lib


# planner switches of uaps_conv_set_tuning (include/uaps_hip.h: UAPS_TUNE_*) <- the environment variables the ablation and
# diagnosis scripts set; the library itself reads no environment
_TUNE_ENV = (("UAPS_DIAG_NO_SPLIT_FWD", 1, None), ("UAPS_DIAG_NO_SPLIT_WRW", 2, None), ("UAPS_DIAG_NO_SMALL", 4, None),
             ("UAPS_DIAG_NO_HP16", 8, None), ("UAPS_SWRW_COLMAJOR", 16, "0"), ("UAPS_WRW_TALL", 32, "0"), ("UAPS_FWD_TALL", 64, "0"),
             ("UAPS_DIAG_NO_ROW16", 128, None), ("UAPS_DIAG_NO_ROW_WRW", 256, None), ("UAPS_DIAG_DEEP_ROWS", 512, None),
             ("UAPS_DIAG_G1_NARROW", 1024, None), ("UAPS_DIAG_NO_G", 2048, None), ("UAPS_DIAG_G_DEEP", 4096, None))


def _configure_from_environment(l) -> None:
    mode = config.text("UAPS_CONV_MODE")
    if mode:                                    # "0" / "exact" / "f32";  "1" / "bf16" / "split";  "2" / "h16"
        m = 0 if mode[0] in "0ef" else (1 if mode[0] in "1bs" else 2)
        if l.uaps_conv_set_mode(m) != 0:
            raise UapsHipError(f"UAPS_CONV_MODE={mode}: uaps_conv_set_mode failed")
    flags = 0
    for name, bit, on_value in _TUNE_ENV:
        v = config.text(name)
        if v is not None and (on_value is None or v == on_value):
            flags |= bit
    l.uaps_conv_set_tuning(flags)


class quiet_gc:
    """Around a hipGraph capture: collect garbage first and keep the cyclic collector off while the stream captures.  The step's
    capture runs thousands of lines of Python; a collection in the middle of it finalises whatever dead cycles the process holds --
    other trainers' CUDAGraph objects (hipGraphExecDestroy), tensors of other private pools -- i.e. runtime calls from inside a
    global-mode capture.  torch.cuda.graph() used to collect before capturing and no longer does (torch.compiler.config.
    force_cudagraph_gc, off by default); seen as an intermittent segmentation fault in capture_end in a test process that had
    built and dropped a dozen captured trainers."""

    def __enter__(self):
        import gc
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False


class WrwReduceItem(C.Structure):
    """uaps_wrw_reduce_item (include/uaps_hip.h)."""
    _fields_ = [("workspace", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p), ("B", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
                ("H", C.c_int), ("W", C.c_int), ("ks", C.c_int), ("cfg", C.c_int)]


class CallHints(C.Structure):
    """uaps_call_hints (include/uaps_hip.h): the optional operands of ONE call, handed to the *_h entry points as their first argument."""
    _fields_ = [("struct_size", C.c_uint), ("bound", C.c_void_p * 3), ("mul", C.c_float * 3), ("out_amax", C.c_void_p),
                ("stats_mean", C.c_void_p), ("stats_bias", C.c_void_p), ("residual", C.c_void_p),
                ("dyt_y", C.c_void_p), ("dyt_coef", C.c_void_p), ("dyt_out", C.c_void_p), ("dyt_slope", C.c_float), ("dyt_groups", C.c_int),
                ("bsum_y", C.c_void_p), ("bsum_mean", C.c_void_p), ("bsum_invstd", C.c_void_p), ("bsum_gamma", C.c_void_p),
                ("bsum_beta", C.c_void_p), ("bsum_partials", C.c_void_p), ("bsum_max", C.c_void_p), ("bsum_slope", C.c_float),
                ("bsum_groups", C.c_int)]


class ConvCall(C.Structure):
    """uaps_conv_call (include/uaps_hip.h): the explicit, size-versioned form of the convolution entry points (uaps_conv_ex)."""
    _fields_ = [("struct_size", C.c_uint), ("op", C.c_int), ("B", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("H", C.c_int),
                ("W", C.c_int), ("ks", C.c_int), ("cfg", C.c_int), ("x", C.c_void_p), ("C1", C.c_int), ("x2", C.c_void_p),
                ("w_packed", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("y2", C.c_void_p), ("y_grad", C.c_void_p),
                ("stats", C.c_void_p), ("xf", C.c_void_p), ("xf_slope", C.c_float), ("xf_groups", C.c_int), ("want_bias", C.c_int),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p), ("hints", CallHints)]


def mk_hints(bounds=(), out_amax=None, stats=None, residual=None, dyt=None, bsum=None):
    """The uaps_call_hints argument of a *_h entry point (a ctypes byref; pass None for "no hints").
    bounds: up to three (bound tensor, host factor) pairs or None; out_amax: a zeroed bound tensor; stats: (running_mean or
    None, conv bias or None) = the per-channel shift BatchNorm partial sums are formed about / were formed about; residual: the
    tensor a BatchNorm apply pass adds before its ReLU (residual joins).
    The record is an ARGUMENT: the library copies it inside the call it is passed to and keeps nothing.  The buffer behind the returned
    reference is this Python thread's and is rewritten by the next mk_hints of the thread, so build it in the call expression."""
    h = getattr(_tls, "hints", None)
    if h is None:
        h = _tls.hints = CallHints()
        h.struct_size = C.sizeof(CallHints)
    for i in range(3):
        h.bound[i] = None
    h.out_amax = h.stats_mean = h.stats_bias = None
    h.residual = residual.data_ptr() if residual is not None else None
    if dyt is not None:                       # (y, coef, out, slope, groups): uaps_call_hints::dyt_* (uaps_bn_act_bwd_prepare)
        h.dyt_y, h.dyt_coef, h.dyt_out, h.dyt_slope, h.dyt_groups = dyt[0].data_ptr(), dyt[1].data_ptr(), dyt[2].data_ptr(), float(dyt[3]), int(dyt[4])
    else:
        h.dyt_y = h.dyt_coef = h.dyt_out = None
        h.dyt_slope, h.dyt_groups = 0.0, 0
    if bsum is not None:                      # (y, mean, invstd, gamma, beta, partials, maxes, slope, groups): uaps_call_hints::bsum_*
        (h.bsum_y, h.bsum_mean, h.bsum_invstd, h.bsum_gamma, h.bsum_beta, h.bsum_partials, h.bsum_max) = [t.data_ptr() for t in bsum[:7]]
        h.bsum_slope, h.bsum_groups = float(bsum[7]), int(bsum[8])
    else:
        h.bsum_y = h.bsum_mean = h.bsum_invstd = h.bsum_gamma = h.bsum_beta = h.bsum_partials = h.bsum_max = None
        h.bsum_slope, h.bsum_groups = 0.0, 0
    if stats is not None:
        if stats[0] is not None:
            h.stats_mean = stats[0].data_ptr()
        if stats[1] is not None:
            h.stats_bias = stats[1].data_ptr()
    for i, b in enumerate(bounds):
        if b is not None:
            h.bound[i], h.mul[i] = b[0].data_ptr(), float(b[1])
    if out_amax is not None:
        h.out_amax = out_amax.data_ptr()
    return C.byref(h)


class LaunchTimer:
    """Times the MAIN kernel launched by the entry point(s) called inside the `with` block: the two events are attached to
    that kernel's dispatch (uaps_next_launch_events), so `elapsed_ms()` after a synchronize is the kernel's execution time
    as rocprofv3's kernel trace reports it.  When no main launch consumed them (an entry point without one), they bracket
    the block on the stream instead (event-to-event: includes the launch gaps)."""

    def __enter__(self):
        import torch
        self.start, self.stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.start.record()                 # creates the hipEvent_t; also the start mark of the fall-back
        self.stop.record()
        lib().uaps_next_launch_events(self.start.cuda_event, self.stop.cuda_event)
        return self

    def __exit__(self, *a):
        self.dispatch = bool(lib().uaps_next_launch_events(None, None))
        if not self.dispatch:
            self.stop.record()
        return False

    def elapsed_ms(self) -> float:
        return self.start.elapsed_time(self.stop)


_error_words = {}      # device index -> int32[1] tensor, never freed (the library holds its raw pointer)


def error_word(device):
    """The sticky error word of `device` (include/uaps_hip.h, uaps_set_error_word): allocated once per device, bound for good and
    never freed, so that the raw pointer the library keeps cannot dangle whatever happens to trainers and models.  Shared by
    everything that runs on the device; UAPSTrainer.check_errors reads and clears it."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    w = _error_words.get(idx)
    if w is None:
        w = _error_words[idx] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
    with torch.cuda.device(idx):           # (re)bound on every call: a caller of the C ABI may have pointed the library elsewhere
        check(lib().uaps_set_error_word(w.data_ptr()), "uaps_set_error_word")
    return w


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().uaps_error_string(rc).decode()
        raise UapsHipError(f"{what} failed: {msg} (code {rc})")


def ptr_array(tensors):
    """Host array of device pointers (const float* const*)."""
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def current_stream(device=None) -> int:
    """Raw hipStream_t of torch's current stream on `device` (fast path: no Stream object is built)."""
    import torch
    if device is None:
        idx = torch.cuda.current_device()
    else:
        idx = device.index if isinstance(device, torch.device) else int(device)
        if idx is None:
            idx = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)


class device_guard:
    """`with device_guard(dev):` = torch.cuda.device(dev), skipped when dev already is the current device (the usual
    case: one process per GPU), which saves a few microseconds on each of the ~700 launches of a step."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        import torch
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        self.ctx = None if idx == torch.cuda.current_device() else torch.cuda.device(idx)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.ctx is not None:
            return self.ctx.__exit__(*a)
        return False


def require_device(t, what: str):
    """The kernels only exist for the GPU: refuse CPU tensors loudly instead of computing elsewhere."""
    if not t.is_cuda:
        raise UapsHipError(f"{what}: tensor is on {t.device}; uaps_amd ops run only on a ROCm device "
                           "(hand-written HIP kernels, no CPU fallback)")
