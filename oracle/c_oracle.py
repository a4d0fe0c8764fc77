"""ctypes binding of oracle/uaps_loss_ref.c.  TEST INFRASTRUCTURE, NOT PRODUCT (see uaps_oracle.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libuaps_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "uaps_loss_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"] if force else ["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.uaps_ref_unsup_nstats.restype = C.c_size_t
        _lib.uaps_ref_sup_nstats.restype = C.c_size_t
    return _lib


def _ptrs(arrs, ctype=C.c_float):
    P = C.POINTER(ctype)
    return (P * len(arrs))(*[a.ctypes.data_as(P) for a in arrs])


def _f32(a):
    return [np.ascontiguousarray(x, dtype=np.float32) for x in a]


def unsup_fwd(logits, w):
    """logits: sequence of D float32 [B,C,H,W]; w: D float64. Returns dict."""
    L = lib(); z = _f32(logits); D = len(z); B, Cc, H, W = z[0].shape
    pseudo = np.empty((B, H, W), np.int64); var = np.empty((D, B, H, W), np.float32)
    mixed = np.empty((B, Cc, H, W), np.float32)
    stats = np.zeros(L.uaps_ref_unsup_nstats(D, Cc), np.float64)
    w = np.ascontiguousarray(w, np.float64)
    rc = L.uaps_ref_unsup_fwd(_ptrs(z), w.ctypes.data_as(C.POINTER(C.c_double)), D, B, Cc, H, W,
                              pseudo.ctypes.data_as(C.POINTER(C.c_int64)), var.ctypes.data_as(C.POINTER(C.c_float)),
                              mixed.ctypes.data_as(C.POINTER(C.c_float)), stats.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0, rc
    return {"pseudo": pseudo, "var": var, "mixed": mixed, "stats": stats}


def unsup_losses(stats, D, Cc, N, eps=1e-7):
    out = np.zeros(4 * D + 2, np.float64)
    lib().uaps_ref_unsup_losses(stats.ctypes.data_as(C.POINTER(C.c_double)), D, Cc, C.c_long(N), C.c_double(eps),
                                out.ctypes.data_as(C.POINTER(C.c_double)))
    return {"ce": out[:D], "dice": out[D:2 * D], "s": out[2 * D:3 * D], "E": out[3 * D:4 * D],
            "ps_loss": out[4 * D], "l_uncert": out[4 * D + 1]}


def unsup_bwd(logits, pseudo, stats, cw1, cw2, gscale=1.0, eps=1e-7):
    L = lib(); z = _f32(logits); D = len(z); B, Cc, H, W = z[0].shape
    g = [np.empty_like(a) for a in z]
    pseudo = np.ascontiguousarray(pseudo, np.int64)
    rc = L.uaps_ref_unsup_bwd(_ptrs(z), pseudo.ctypes.data_as(C.POINTER(C.c_int64)),
                              stats.ctypes.data_as(C.POINTER(C.c_double)), C.c_double(cw1), C.c_double(cw2),
                              C.c_double(gscale), C.c_double(eps), D, B, Cc, H, W, _ptrs(g))
    assert rc == 0, rc
    return g


def sup_fwd(logits, labels):
    L = lib(); z = _f32(logits); D = len(z); B, Cc, H, W = z[0].shape
    labels = np.ascontiguousarray(labels, np.int64)
    stats = np.zeros(L.uaps_ref_sup_nstats(D, Cc), np.float64)
    rc = L.uaps_ref_sup_fwd(_ptrs(z), labels.ctypes.data_as(C.POINTER(C.c_int64)), D, B, Cc, H, W,
                            stats.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0, rc
    return stats


def sup_losses(stats, D, Cc, N, eps=1e-7):
    out = np.zeros(2 * D + 1, np.float64)
    lib().uaps_ref_sup_losses(stats.ctypes.data_as(C.POINTER(C.c_double)), D, Cc, C.c_long(N), C.c_double(eps),
                              out.ctypes.data_as(C.POINTER(C.c_double)))
    return {"ce": out[:D], "dice": out[D:2 * D], "sup": out[2 * D]}


def sup_bwd(logits, labels, stats, gscale=1.0, eps=1e-7):
    L = lib(); z = _f32(logits); D = len(z); B, Cc, H, W = z[0].shape
    g = [np.empty_like(a) for a in z]
    labels = np.ascontiguousarray(labels, np.int64)
    rc = L.uaps_ref_sup_bwd(_ptrs(z), labels.ctypes.data_as(C.POINTER(C.c_int64)),
                            stats.ctypes.data_as(C.POINTER(C.c_double)), C.c_double(gscale), C.c_double(eps),
                            D, B, Cc, H, W, _ptrs(g))
    assert rc == 0, rc
    return g


def confusion(logits, labels):
    z = np.ascontiguousarray(logits, np.float32); B, Cc, H, W = z.shape
    labels = np.ascontiguousarray(labels, np.int64)
    counts = np.zeros((Cc, Cc), np.int64)
    rc = lib().uaps_ref_confusion(z.ctypes.data_as(C.POINTER(C.c_float)), labels.ctypes.data_as(C.POINTER(C.c_int64)),
                                  B, Cc, H, W, counts.ctypes.data_as(C.POINTER(C.c_int64)))
    assert rc == 0, rc
    return counts
