"""Evaluation path of the reference (UAPS-Testing.ipynb, the only consumer of the checkpoints UAPS_train.py writes):
main-head prediction, the all-heads ensemble of the paper's decoder study, the test-time uncertainty map, and the
per-image score table.  Everything runs on the HIP kernels of this package; nothing here trains."""
from __future__ import annotations

import operator
from typing import Dict, Iterable, Optional, Tuple

import numpy as np
import torch

from . import consistency, losses, metrics


@torch.no_grad()
def predict(model: torch.nn.Module, images: torch.Tensor) -> Tuple[torch.Tensor, Tuple[torch.Tensor, ...]]:
    """model.eval(); main-head arg-max mask [B,H,W] and the raw head logits (notebook cells 11-13: `output, ax1, _, _ =
    model(image); masked = torch.argmax(output, dim=1)`).  As in the reference, the auxiliary decoders stay perturbed
    in eval mode (Dropout is called with training=True unconditionally, UAPS_unet.py:156-158)."""
    model.eval()
    out = model(images)
    heads = out if isinstance(out, (tuple, list)) else (out,)
    return torch.argmax(heads[0], dim=1), tuple(heads)


@torch.no_grad()
def predict_main(model: torch.nn.Module, images: torch.Tensor) -> torch.Tensor:
    """Main-head arg-max mask with ONLY the encoder and the main decoder run (the paper's "main decoder only" inference row,
    fig_data/decoder-effect.jpg: its time does not depend on the number of auxiliary decoders); the same mask as predict()[0].
    Models without a `forward_main` (the ResNet variant) run every head."""
    model.eval()
    fm = getattr(model, "forward_main", None)
    logits = fm(images) if fm is not None else predict(model, images)[1][0]
    return torch.argmax(logits, dim=1)


_VERSION = operator.attrgetter("_version")


class CapturedMainHead:
    """predict_main for ONE input shape as a replayed hipGraph: at batch 1 the eager call is ~90 launches of a few microseconds of
    work each and is bound by their issue; the replay hands the device the whole chain at once.  Same kernels and arithmetic as
    predict_main (the main decoder has no perturbation: nothing random is baked in).  The packed weights are read through the
    graph's pointers: build a new object after the model's parameters changed (a checkpoint load, a training step).

        run = CapturedMainHead(model, example_images)
        mask = run(images)            # int64 [B,H,W], valid until the next call (clone() it to keep it)
    """

    def __init__(self, model: torch.nn.Module, example: torch.Tensor, warmup: int = 2):
        from . import bounds
        if not example.is_cuda:
            raise ValueError("CapturedMainHead: a GPU tensor expected")
        self.model, self.x = model, example.detach().clone()
        self._tensors = self._params = None
        model.eval()
        cur = torch.cuda.current_stream(example.device)
        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():       # warm-up on a side stream (torch's capture recipe): packs the weights, sizes the workspaces
            for _ in range(max(1, warmup)):
                predict_main(model, self.x)
        cur.wait_stream(side)
        torch.cuda.synchronize(example.device)
        bounds.reset_pool()                                  # the zero fill of every max|.| scalar the kernels raise belongs to the graph
        self.graph = torch.cuda.CUDAGraph()
        from . import _lib
        # thread-local error mode: another thread's runtime calls (a DataLoader's pin thread, a watchdog) do not invalidate the capture;
        # quiet_gc: no finaliser of THIS thread runs inside it
        with _lib.quiet_gc(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"), torch.no_grad():
            self.out = predict_main(model, self.x)
        bounds.reset_pool()                                  # eager code must not be handed scalars the replays re-zero
        # the graph reads the packed weights and the BatchNorm buffers through raw addresses: hold the packed buffers (the cache may
        # drop them), and remember what the parameters looked like -- a replay after they changed would silently use stale weights
        from . import conv
        self._held = [(e[3], e[4]) for e in conv._packed.values()]
        self._stamp = self._weights_stamp()

    def _weights_stamp(self):
        """(manual-invalidation generation, number of tensors, sum of the version counters of every parameter and buffer, sum of the
        parameters' optimizer counters).  Every term only ever grows, so a sum changes exactly when one of its terms does.  This runs in
        front of EVERY replay (the tuple-of-tuples form of the first version cost 0.5 ms per call, more than the graph): the versions
        are summed by a C-level loop over a cached list, the optimizer counters are only walked when some optimizer of the process has
        stepped since the last look (conv._shared_cell)."""
        from . import conv
        if self._tensors is None:
            # a model with forward_main runs its encoder and main decoder only (unet.UNet_UAPS): watch those tensors, a third of the net
            mods = [getattr(self.model, n, None) for n in ("encoder", "main_decoder")] if hasattr(self.model, "forward_main") else []
            mods = mods if mods and all(isinstance(m, torch.nn.Module) for m in mods) else [self.model]
            self._params = [p for m in mods for p in m.parameters()]
            self._tensors = self._params + [b for m in mods for b in m.buffers()]
            self._seen_steps, self._cell_sum = -1, 0
        if conv._shared_cell[0] != self._seen_steps:
            self._seen_steps = conv._shared_cell[0]
            self._cell_sum = sum(c[0] for c in (getattr(p, "_uaps_cell", None) for p in self._params) if c is not None)
        return (conv._generation, len(self._tensors), sum(map(_VERSION, self._tensors)), self._cell_sum)

    def stale(self) -> bool:
        """Have the model's parameters or buffers changed since the capture (an optimizer step, a checkpoint load, an in-place edit)?"""
        return self._weights_stamp() != self._stamp

    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        if images.shape != self.x.shape or images.device != self.x.device or images.dtype != self.x.dtype:
            raise ValueError(f"CapturedMainHead: captured for {tuple(self.x.shape)} {self.x.dtype} on {self.x.device}")
        if self.stale():
            raise RuntimeError("CapturedMainHead: the model's parameters or buffers changed since the capture (optimizer step, checkpoint "
                               "load, in-place edit): the graph holds the OLD packed weights -- build a new CapturedMainHead")
        self.x.copy_(images, non_blocking=True)
        self.graph.replay()
        return self.out


@torch.no_grad()
def ensemble_from_heads(heads) -> torch.Tensor:
    """Arg-max of the mean softmax of the given head logits: `uaps_unsup_fwd` (csrc/loss_kernels.hpp) with uniform mixing weights --
    its pseudo-label IS argmax_c sum_k w_k softmax(z_k)_c (UAPS_train.py:251-255) -- one pass over the D logit tensors instead of D
    softmax launches and D - 1 adds.  int64 [B,H,W]."""
    D = len(heads)
    if D == 1:
        return torch.argmax(heads[0], dim=1)
    return losses.uaps_unsup_loss(tuple(heads), [1.0 / D] * D, 0.0, 0.0).pseudo


@torch.no_grad()
def predict_ensemble(model: torch.nn.Module, images: torch.Tensor) -> torch.Tensor:
    """Arg-max of the mean softmax over all heads (the "ensemble" rows of the paper's decoder study, README.md:107-111; notebook
    cells 11-19 with every head's softmax averaged), on the loss block's mixing kernel."""
    _, heads = predict(model, images)
    return ensemble_from_heads(heads)


@torch.no_grad()
def predict_with_uncertainty(model: torch.nn.Module, images: torch.Tensor):
    """(mask, uncertainty, confidence): notebook cell 24 -- uncertainty = sum_c KLDivLoss('none')(log_softmax(main),
    softmax(aux1)) per pixel, confidence = 1 - uncertainty."""
    mask, heads = predict(model, images)
    if len(heads) < 2:
        raise ValueError("the uncertainty map needs an auxiliary head")
    var1 = consistency.uncertainty_map(heads[0], heads[1])
    return mask, var1, 1.0 - var1


@torch.no_grad()
def evaluate(model: torch.nn.Module, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]]) -> Dict[str, float]:
    """The score table of notebook cell 25: mIoU, mDice and pixel accuracy of the main head computed image by image
    (utilities/metrics.py conventions) and averaged over the images, in percent."""
    miou, mdice, acc = [], [], []
    for x, y in batches:
        _, heads = predict(model, x)
        cms = metrics.seg_confusion_per_image(heads[0], y).cpu().numpy()
        for cm in cms:
            m = metrics.metrics_from_confusion(cm)
            miou.append(m["miou"]); mdice.append(m["mdice"]); acc.append(m["acc"])
    return {"mIoU(%)": float(np.mean(miou) * 100), "mDice(%)": float(np.mean(mdice) * 100), "Accuracy(%)": float(np.mean(acc) * 100),
            "images": len(acc)}


def load_for_inference(model: torch.nn.Module, checkpoint_path: str, device: Optional[torch.device] = None) -> Dict:
    """Notebook cell 4: `checkpoint = torch.load(path); model.load_state_dict(checkpoint['state_dict'])` -- accepts the
    `module.`-prefixed keys of the reference's nn.DataParallel checkpoints (UAPS_train.py:443-450) and plain ones."""
    from .trainer import load_state_dict_any_prefix
    ck = torch.load(checkpoint_path, map_location=device or "cpu", weights_only=False)
    load_state_dict_any_prefix(model, ck["state_dict"] if "state_dict" in ck else ck)
    return ck
