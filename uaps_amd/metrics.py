"""Segmentation metrics of the reference (utilities/metrics.py:8-61) from one on-device confusion
matrix (csrc/perturb.hip `uaps_seg_confusion`): one kernel and, when a number is wanted, one 8..512-byte
device->host copy instead of the reference's ~9 `.item()` syncs per call."""
from __future__ import annotations

import warnings

import numpy as np
import torch

from . import _lib


def seg_confusion(logits: torch.Tensor, labels: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """counts[t, p] = #pixels with label t whose arg-max prediction is p (int64 [C,C], on device).
    argmax(softmax(z)) == argmax(z), first maximum wins, as torch.argmax in metrics.py:10,19,43."""
    _lib.require_device(logits, "seg_confusion")
    if logits.dtype != torch.float32 or logits.dim() != 4:
        raise TypeError("logits must be float32 [B,C,H,W]")
    B, C, H, W = logits.shape
    if labels.shape != (B, H, W):
        raise ValueError("labels must be [B,H,W]")
    z = logits.detach().contiguous()
    y = labels.to(torch.int64).contiguous()
    counts = out if out is not None else torch.empty((C, C), dtype=torch.int64, device=z.device)
    with _lib.device_guard(z.device):
        rc = _lib.lib().uaps_seg_confusion(z.data_ptr(), y.data_ptr(), B, C, H, W, counts.data_ptr(),
                                           _lib.current_stream(z.device))
    _lib.check(rc, "uaps_seg_confusion")
    return counts


def seg_confusion_per_image(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """One confusion matrix per image, int64 [B,C,C] (the evaluation notebooks score image by image and average the
    scores, UAPS-Testing.ipynb cells 11-19, 25)."""
    _lib.require_device(logits, "seg_confusion_per_image")
    if logits.dtype != torch.float32 or logits.dim() != 4:
        raise TypeError("logits must be float32 [B,C,H,W]")
    B, C, H, W = logits.shape
    if labels.shape != (B, H, W):
        raise ValueError("labels must be [B,H,W]")
    z = logits.detach().contiguous()
    y = labels.to(torch.int64).contiguous()
    counts = torch.empty((B, C, C), dtype=torch.int64, device=z.device)
    L = _lib.lib()
    with _lib.device_guard(z.device):
        st = _lib.current_stream(z.device)
        for b in range(B):
            rc = L.uaps_seg_confusion(z.data_ptr() + 4 * b * C * H * W, y.data_ptr() + 8 * b * H * W, 1, C, H, W,
                                      counts.data_ptr() + 8 * b * C * C, st)
            _lib.check(rc, "uaps_seg_confusion")
    return counts


def metrics_from_confusion(cm, smooth: float = 1e-10):
    """{'miou','mdice','acc'} with the reference's conventions: classes 1..C-1 only, NaN for classes
    without ground-truth pixels, nanmean over classes (metrics.py:23-37, 47-61); accuracy over all
    pixels (metrics.py:8-13).  `cm` is a host array or a device tensor (copied once)."""
    if isinstance(cm, torch.Tensor):
        cm = cm.cpu().numpy()
    cm = np.asarray(cm, dtype=np.float64)
    C = cm.shape[0]
    ious, dices = [], []
    for c in range(1, C):
        n_lab, n_pred, inter = cm[c, :].sum(), cm[:, c].sum(), cm[c, c]
        if n_lab == 0:
            ious.append(np.nan)
            dices.append(np.nan)
        else:
            union = n_lab + n_pred - inter
            ious.append((inter + smooth) / (union + smooth))
            dices.append(2 * (inter + smooth) / (union + inter + smooth))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        miou = float(np.nanmean(ious)) if ious else float("nan")
        mdice = float(np.nanmean(dices)) if dices else float("nan")
    tot = cm.sum()
    return {"miou": miou, "mdice": mdice, "acc": float(np.trace(cm) / tot) if tot else float("nan")}


def mean_batch_metrics(cms, smooth: float = 1e-10):
    """What the reference's training / validation loops log (UAPS_train.py:305-306, 320-321, 388-399): mIoU / mDice /
    accuracy of EACH batch (NaN-mean over the classes present in that batch), arithmetically averaged over the batches (a
    batch whose classes 1..C-1 are all absent contributes NaN, as `running += mDice(...)` does).  `cms`: [N, C, C]."""
    if isinstance(cms, torch.Tensor):
        cms = cms.cpu().numpy()
    per = [metrics_from_confusion(c, smooth) for c in cms]
    n = len(per)
    return {k: float(sum(p[k] for p in per) / n) if n else float("nan") for k in ("miou", "mdice", "acc")}


def pixel_accuracy(output, mask):
    """utilities/metrics.py:8-13."""
    return metrics_from_confusion(seg_confusion(output, mask))["acc"]


def mIoU(pred_mask, mask, smooth=1e-10, n_classes=None):
    """utilities/metrics.py:16-37 (n_classes defaults to the logit channel count instead of a literal 4)."""
    return metrics_from_confusion(seg_confusion(pred_mask, mask), smooth)["miou"]


def mDice(pred_mask, mask, smooth=1e-10, n_classes=None):
    """utilities/metrics.py:40-61."""
    return metrics_from_confusion(seg_confusion(pred_mask, mask), smooth)["mdice"]
