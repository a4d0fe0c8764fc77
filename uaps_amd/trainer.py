"""The UAPS training step and its optimizer / checkpoint surface (reference UAPS_train.py:109-450).

One `train_step` = the body of the reference's iteration loop (UAPS_train.py:177-306): two forwards
of the multi-decoder net in train mode (labelled, then unlabelled batch: BatchNorm statistics are
per forward), the fused HIP loss block, backward, (RCCL gradient average), Adam, and the running
confusion matrix for mIoU/mDice -- with no device->host synchronisation inside the step (the
reference does ~27 `.item()` syncs per step, UAPS_train.py:295-306).
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Optional

import numpy as np
import torch

from . import dist as udist
from . import losses, metrics, optim, perturb
from .ramps import get_current_consistency_weight


class UAPSTrainer:
    def __init__(self, model: torch.nn.Module, base_lr: float = 1e-3, consistency1: float = 0.1,
                 consistency2: float = 0.1, consistency_rampup: float = 200, ramp_divisor: int = 80,
                 seed: int = 1337, loss_fn: Optional[Callable] = None, overlap_comm: bool = True,
                 track_metrics: bool = True, pair_forward: bool = True):
        self.model = model
        params = list(model.parameters())
        self.device = params[0].device
        on_gpu = self.device.type == "cuda"
        # UAPS_train.py:112-113
        # same class contract and state_dict as torch.optim.Adam; on the GPU the step is one hand-written multi-tensor kernel
        self.optimizer = optim.Adam(params, lr=base_lr) if on_gpu else torch.optim.Adam(params, lr=base_lr)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="max", min_lr=1e-8, patience=50)
        self.c1, self.c2, self.rampup, self.ramp_divisor = consistency1, consistency2, consistency_rampup, ramp_divisor
        self.iter_num = 0
        self.n_heads = len(list(model.aux_decoders())) + 1 if hasattr(model, "aux_decoders") else 1
        # the Dirichlet mixing weights must be the same on every rank: a private, identically seeded stream
        self.mix_rng = np.random.RandomState(seed)
        self.rank, self.world = udist.rank(), udist.world_size()
        np.random.seed(seed + self.rank)                      # FeatureDropout thresholds (numpy global RNG)
        perturb.manual_seed(seed, self.rank)                  # Philox streams offset by rank
        self.loss_fn = loss_fn or losses.uaps_step_loss       # no fallback: raises without the HIP library
        # one pass over the concatenated labelled+unlabelled batch (per-half BatchNorm statistics and perturbation
        # draws keep the two-forward semantics) instead of two forwards; needs the HIP kernels, i.e. a GPU model
        self.pair_forward = pair_forward and on_gpu and loss_fn is None and hasattr(model, "forward_pair")
        self.track_metrics = track_metrics and loss_fn is None
        self.buckets = udist.GradBuckets(model, overlap=overlap_comm) if self.world > 1 else None
        self.confusion = None
        self.last: Dict[str, torch.Tensor] = {}

    # -- schedule (UAPS_train.py:279-280) --
    def consistency_weights(self):
        return (get_current_consistency_weight(self.c1, self.iter_num, self.rampup, self.ramp_divisor),
                get_current_consistency_weight(self.c2, self.iter_num, self.rampup, self.ramp_divisor))

    def train_step(self, x_l: torch.Tensor, y_l: torch.Tensor, x_u: torch.Tensor, w=None) -> Dict[str, torch.Tensor]:
        """Returns device scalars (loss, sup, unsup); nothing here synchronises with the host."""
        if not self.model.training:                      # model.train() walks ~2600 modules: only when the mode changes
            self.model.train()
        cw1, cw2 = self.consistency_weights()
        if self.pair_forward and x_l.shape == x_u.shape:
            both = self.model.forward_pair(x_l, x_u)                              # UAPS_train.py:177 + :185 in one pass
            if w is None:
                w = self.mix_rng.dirichlet(np.ones(len(both)), size=1)[0]        # :251
            out = losses.uaps_pair_loss(both, y_l, w, cw1, cw2)                   # :186-282
            lab = tuple(z[: x_l.shape[0]] for z in both)
        else:
            lab = self.model(x_l)                                                 # UAPS_train.py:177
            un = self.model(x_u)                                                  # :185
            if not isinstance(lab, (tuple, list)):
                lab, un = (lab,), (un,)
            if w is None:
                w = self.mix_rng.dirichlet(np.ones(len(un)), size=1)[0]          # :251
            out = self.loss_fn(lab, y_l, un, w, cw1, cw2)                         # :186-282
        self.optimizer.zero_grad(set_to_none=True)                                # :285
        out.loss.backward()                                                       # :287
        if self.buckets is not None:
            self.buckets.finish()
        self.optimizer.step()                                                     # :292
        if self.track_metrics:                                                    # :305-306 (main head, labelled batch)
            cm = metrics.seg_confusion(lab[0], y_l)
            self.confusion = cm if self.confusion is None else self.confusion + cm
        self.iter_num += 1
        self.last = {"loss": out.loss.detach(), "sup": out.sup.detach(), "unsup": out.unsup.detach(),
                     "cw1": cw1, "cw2": cw2, "w": w}
        return self.last

    def epoch_metrics(self, reset: bool = True) -> Dict[str, float]:
        """mIoU / mDice / accuracy over the labelled batches seen since the last reset (one D2H copy)."""
        if self.confusion is None:
            return {"miou": float("nan"), "mdice": float("nan"), "acc": float("nan")}
        m = metrics.metrics_from_confusion(self.confusion)
        if reset:
            self.confusion = None
        return m

    @torch.no_grad()
    def validate(self, batches) -> Dict[str, float]:
        """UAPS_train.py:367-393: eval mode, main head only, CE + (1 - mDice), mIoU, accuracy."""
        self.model.eval()
        cm, ce_sum, n = None, None, 0
        for x, y in batches:
            out = self.model(x)
            main = out[0] if isinstance(out, (tuple, list)) else out
            ce = losses.ce_loss(main, y)
            ce_sum = ce if ce_sum is None else ce_sum + ce
            c = metrics.seg_confusion(main, y)
            cm = c if cm is None else cm + c
            n += 1
        m = metrics.metrics_from_confusion(cm)
        m["ce"] = float(ce_sum) / max(n, 1)
        m["loss"] = 0.5 * ((1 - m["mdice"]) + m["ce"])
        return m

    # -- checkpoint (UAPS_train.py:437-450) --
    def state_for_checkpoint(self, epoch: int, best_dice: float, dataparallel_prefix: bool = True) -> Dict:
        sd = self.model.state_dict()
        if dataparallel_prefix and not any(k.startswith("module.") for k in sd):
            sd = {"module." + k: v for k, v in sd.items()}     # the reference saves nn.DataParallel(model).state_dict()
        return {"epoch": epoch, "best_dice_1": best_dice, "state_dict": sd, "optimizer": self.optimizer.state_dict()}

    def save_checkpoint(self, path: str, epoch: int, best_dice: float):
        if self.rank == 0:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            torch.save(self.state_for_checkpoint(epoch, best_dice), path)

    def load_checkpoint(self, path: str, load_optimizer: bool = True) -> Dict:
        ck = torch.load(path, map_location=self.device, weights_only=False)
        load_state_dict_any_prefix(self.model, ck["state_dict"])
        if load_optimizer and "optimizer" in ck:
            self.optimizer.load_state_dict(ck["optimizer"])
        return ck


def load_state_dict_any_prefix(model: torch.nn.Module, sd: Dict[str, torch.Tensor]):
    """Accepts the reference's `module.`-prefixed checkpoints (UAPS_model.py:13) and un-prefixed ones."""
    want_prefix = any(k.startswith("module.") for k in model.state_dict())
    has_prefix = any(k.startswith("module.") for k in sd)
    if has_prefix and not want_prefix:
        sd = {k[len("module."):]: v for k, v in sd.items()}
    elif want_prefix and not has_prefix:
        sd = {"module." + k: v for k, v in sd.items()}
    return model.load_state_dict(sd)
