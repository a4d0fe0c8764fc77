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

from . import _lib, config, conv, lazybn
from . import dist as udist
from . import losses, metrics, optim, perturb
from .ramps import get_current_consistency_weight

_EARLY_ADAM = config.flag("UAPS_EARLY_ADAM", True)      # the decoders' Adam step beside the encoder's backward (world size 1)


class UAPSTrainer:
    def __init__(self, model: torch.nn.Module, base_lr: float = 1e-3, consistency1: float = 0.1,
                 consistency2: float = 0.1, consistency_rampup: float = 200, ramp_divisor: int = 80,
                 seed: int = 1337, loss_fn: Optional[Callable] = None, overlap_comm: bool = True,
                 track_metrics: bool = True, pair_forward: bool = True, gathered_loss: bool = False, step_state: bool = False,
                 use_graph: bool = False):
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
        self._one = None                                 # the unit gradient handed to loss.backward (_unit_gradient)
        self._late_params = None                         # model.decoder_parameters() (_early_adam), asked at the first step
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
        # gathered_loss: CE means / Dice sums / uncertainty means over the batch of ALL ranks, as the reference's
        # nn.DataParallel computes them on the gathered logits (UAPS_model.py:13, UAPS_train.py:194-277): one extra
        # latency-only all-reduce of the raw loss sums per step, and the ranks' gradients are added instead of averaged.
        # Default off = standard data parallelism (every rank's loss is normalised over its own shard).
        self.gathered_loss = bool(gathered_loss) and self.world > 1
        self.exchange = udist.exchange_sums() if self.gathered_loss else None
        self.buckets = udist.GradBuckets(model, overlap=overlap_comm, average=not self.gathered_loss) if self.world > 1 else None
        self._cms = []                                        # one on-device C x C confusion matrix per training step
        self.last: Dict[str, torch.Tensor] = {}
        # sticky device error word (include/uaps_hip.h, uaps_set_error_word): the fp16-split convolutions flag non-finite
        # outputs there -- a violated magnitude bound, or non-finite data -- and check_errors() raises on it
        # one never-freed word per device, bound once (_lib.error_word): no pointer of a dropped trainer is left in the library
        self._err = _lib.error_word(self.device) if on_gpu else None
        # step_state: per-step scalars and the RNG key travel through the device-resident step state instead of kernel
        # arguments (uaps_amd.graph); use_graph: that step is captured as a hipGraph after two warm-up steps and replayed
        self.step_graph = None
        if (step_state or use_graph) and on_gpu:
            from .graph import StepGraph
            self.step_graph = StepGraph(self, capture=bool(use_graph))

    # -- device-side error reporting --
    def check_errors(self, flags: Optional[int] = None) -> None:
        """Raise if a kernel reported an error since the last check (one device->host copy unless `flags` is given)."""
        if self._err is None:
            return
        if flags is None:
            flags = int(self._err.item())
        if flags:
            self._err.zero_()
            what = []
            if flags & 1:
                what.append("a forward / input-gradient convolution in the fp16-split form stored non-finite values")
            if flags & 2:
                what.append("a weight-gradient convolution in the fp16-split form produced non-finite sums")
            raise _lib.UapsHipError(
                "uaps_amd: " + "; ".join(what) + f" (error word {flags:#x}): either non-finite data reached the convolution or a "
                "magnitude bound (tensor attribute _uaps_bound) was smaller than the tensor it describes -- in-place edits of a "
                "bounded tensor, BatchNorm parameters changed between bounds.refresh and the forward, a custom op that kept "
                "the attribute.  conv.set_mode('split') runs without bounds.")

    # -- schedule (UAPS_train.py:279-280) --
    def consistency_weights(self):
        return (get_current_consistency_weight(self.c1, self.iter_num, self.rampup, self.ramp_divisor),
                get_current_consistency_weight(self.c2, self.iter_num, self.rampup, self.ramp_divisor))

    def train_step(self, x_l: torch.Tensor, y_l: torch.Tensor, x_u: torch.Tensor, w=None) -> Dict[str, torch.Tensor]:
        """Returns device scalars (loss, sup, unsup); nothing here synchronises with the host."""
        if self.step_graph is not None and w is None and x_l.shape == x_u.shape:
            return self.step_graph.step(x_l, y_l, x_u)
        if self.step_graph is not None:
            # a step the state-mode body does not cover (caller-supplied mixing weights, a ragged last batch): the plain eager
            # step, with Adam's scalars passed by value for this one step (no step state is active here)
            prev, self.optimizer.from_step_state = self.optimizer.from_step_state, False
            try:
                return self._eager_step(x_l, y_l, x_u, w)
            finally:
                self.optimizer.from_step_state = prev
        return self._eager_step(x_l, y_l, x_u, w)

    def _early_adam(self):
        """deferred_reduces(on_early=...): Adam for the parameters `model.decoder_parameters()` names -- their gradients are final when
        the encoder's part of the backward begins -- on the early flush's side stream; optimizer.step() then takes the rest.
        None (no early step) for data-parallel runs (the exchange comes first), models without that method, foreign optimizers."""
        if not _EARLY_ADAM or self.buckets is not None or not hasattr(self.optimizer, "step_early"):
            return None
        if self._late_params is None:
            m = self.model.module if hasattr(self.model, "module") and not hasattr(self.model, "decoder_parameters") else self.model
            fn = getattr(m, "decoder_parameters", None)           # the model says which gradients are final at the first fan-in
            self._late_params = tuple(fn()) if callable(fn) else ()
        if not self._late_params:
            return None
        if self.optimizer._early_done:
            # the previous step raised between its early and its final Adam call: the decoders' parameters (and their step counts)
            # are one update ahead of the encoder's -- nothing here can undo that, so say it instead of training on
            self.optimizer._early_done.clear()
            raise RuntimeError("uaps_amd: the previous training step failed after the decoders' parameters had been updated (early Adam "
                               "step) and before the encoder's were: the model is half a step out of sync.  Reload the last checkpoint; "
                               "UAPS_EARLY_ADAM=0 keeps the optimizer step in one piece.")
        return lambda: self.optimizer.step_early(self._late_params)

    def _unit_gradient(self, loss: torch.Tensor) -> torch.Tensor:
        """d loss / d loss = 1 as a tensor kept for the trainer's life: `loss.backward()` would fill a fresh one every step (a
        launch in front of the loss backward, on the step's critical path)."""
        one = self._one
        if one is None or one.device != loss.device or one.shape != loss.shape or one.dtype != loss.dtype:
            one = self._one = torch.ones_like(loss)
        return one

    def _eager_step(self, x_l, y_l, x_u, w=None) -> Dict[str, torch.Tensor]:
        if not self.model.training:                      # model.train() walks ~2600 modules: only when the mode changes
            self.model.train()
        cw1, cw2 = self.consistency_weights()
        # forward + backward of a step this trainer drives itself: the two-halves BatchNorm backward may run, and the weight-gradient
        # reductions run batched -- behind the backward, or per bucket in front of its all-reduce (dist.GradBuckets._launch)
        pair = self.pair_forward and x_l.shape == x_u.shape       # (two forwards of one model: autograd sums the two gradients of a weight)
        with lazybn.scope(), conv.deferred_reduces(pair, on_early=self._early_adam(), model=self.model) as step:
            if self.buckets is not None:
                self.buckets.step = step          # the bucket hooks reduce their bucket's deferred weight gradients in front of the all-reduce
            if pair:
                both = self.model.forward_pair(x_l, x_u)                              # UAPS_train.py:177 + :185 in one pass
                if w is None:
                    w = self.mix_rng.dirichlet(np.ones(len(both)), size=1)[0]        # :251
                out = losses.uaps_pair_loss(both, y_l, w, cw1, cw2, exchange=self.exchange)   # :186-282
                lab = tuple(z[: x_l.shape[0]] for z in both)
            else:
                lab = self.model(x_l)                                                 # UAPS_train.py:177
                un = self.model(x_u)                                                  # :185
                if not isinstance(lab, (tuple, list)):
                    lab, un = (lab,), (un,)
                if w is None:
                    w = self.mix_rng.dirichlet(np.ones(len(un)), size=1)[0]          # :251
                if self.exchange is not None:
                    out = self.loss_fn(lab, y_l, un, w, cw1, cw2, exchange=self.exchange)
                else:
                    out = self.loss_fn(lab, y_l, un, w, cw1, cw2)                     # :186-282
            self.optimizer.zero_grad(set_to_none=True)                                # :285
            out.loss.backward(gradient=self._unit_gradient(out.loss))                 # :287
        if self.buckets is not None:
            self.buckets.finish()
        self.optimizer.step()                                                     # :292
        if self.track_metrics:                                                    # :305-306 (main head, labelled batch)
            self._cms.append(metrics.seg_confusion(lab[0], y_l))
        self.iter_num += 1
        self.last = {"loss": out.loss.detach(), "sup": out.sup.detach(), "unsup": out.unsup.detach(),
                     "cw1": cw1, "cw2": cw2, "w": w}
        return self.last

    def epoch_metrics(self, reset: bool = True, pooled: bool = False) -> Dict[str, float]:
        """mIoU / mDice / accuracy of the main head on the labelled batches seen since the last reset, as the reference
        accumulates them: the metric of EACH batch (NaN-mean over the classes present in that batch, utilities/metrics.py:
        16-61) averaged over the batches (UAPS_train.py:305-306, 320-321).  The per-step confusion matrices stay on the
        device; this is the one device->host copy.  `pooled=True` instead scores the summed confusion matrix (not what the
        reference logs: the two differ whenever a class is absent from some batches)."""
        if not self._cms:
            self.check_errors()
            return {"miou": float("nan"), "mdice": float("nan"), "acc": float("nan")}
        if self._err is not None:                    # the error word rides in the same device->host copy
            both = torch.cat([torch.stack(self._cms).flatten(), self._err.to(torch.int64)]).cpu().numpy()
            c = self._cms[0].shape[0]
            cms = both[:-1].reshape(len(self._cms), c, c)
            self.check_errors(int(both[-1]))
        else:
            cms = torch.stack(self._cms).cpu().numpy()
        if reset:
            self._cms = []
        return metrics.metrics_from_confusion(cms.sum(0)) if pooled else metrics.mean_batch_metrics(cms)

    @torch.no_grad()
    def validate(self, batches, pooled: bool = False) -> Dict[str, float]:
        """UAPS_train.py:367-399: eval mode, main head only; per batch CE, 1 - mDice, their half-sum, mIoU, accuracy and
        mDice, each averaged over the batches -- the mDice returned here is what the reference feeds to
        ReduceLROnPlateau.step (:402) and to the best-checkpoint test (:427).  One device->host copy at the end.
        `pooled=True`: metrics of the summed confusion matrix instead (a different number, see epoch_metrics)."""
        self.model.eval()
        cms, ces = [], []
        for x, y in batches:
            out = self.model(x)
            main = out[0] if isinstance(out, (tuple, list)) else out
            ces.append(losses.ce_loss(main, y))
            cms.append(metrics.seg_confusion(main, y))
        if not cms:
            return {"miou": float("nan"), "mdice": float("nan"), "acc": float("nan"), "ce": float("nan"), "loss": float("nan")}
        cm = torch.stack(cms).cpu().numpy()
        ce = torch.stack(ces).double().cpu().numpy()
        self.check_errors()
        if pooled:
            m = metrics.metrics_from_confusion(cm.sum(0))
            m["ce"] = float(ce.mean())
            m["dice_loss"] = 1 - m["mdice"]
            m["loss"] = 0.5 * (m["dice_loss"] + m["ce"])
            return m
        per = [metrics.metrics_from_confusion(c) for c in cm]
        n = len(per)
        m = {k: float(sum(p[k] for p in per) / n) for k in ("miou", "mdice", "acc")}                      # :394-399
        m["ce"] = float(ce.mean())
        m["dice_loss"] = float(sum(1 - p["mdice"] for p in per) / n)                                       # :385, 396
        m["loss"] = float(sum(0.5 * ((1 - p["mdice"]) + c) for p, c in zip(per, ce)) / n)                  # :386, 394
        return m

    # -- the epoch loop (UAPS_train.py:127-159, 316-329, 367-402, 427-450) --
    def fit(self, train_batches, unlabeled_batches, val_batches, epochs: int = 800, iter_per_epoch: int = 60,
            checkpoint_path: Optional[str] = None, start_epoch: int = 1, best_dice: float = 0.0,
            log: Optional[Callable[[Dict], None]] = None):
        """The reference's `Network.run()` around `train_step`, with its off-by-one conventions kept:

        * `for epoch in range(1, epochs)` (:127) -- epochs - 1 epochs; `start_epoch` resumes a run (pass the checkpoint's
          `epoch + 1` and `best_dice_1`, after `load_checkpoint`);
        * every epoch draws `iter_per_epoch - 1` steps (`range(1, iter_per_epoch)`, :159) from a FRESH
          `zip(cycle(train), cycle(unlabeled))` -- the over-sampling form of DAGM-Dataset-codes/UAPS_train.py:143; the NEU
          script's plain `zip` (:157) raises StopIteration as soon as the labelled loader is shorter than an epoch;
        * the logged training means divide the running sums by `iter_per_epoch`, not by the number of steps (:316-321);
        * the consistency ramp runs on `iter_num // 80` across epochs (:279-280; `ramp_divisor`);
        * validation on the main head (:367-399), `scheduler.step(val mDice)` (:402), and the checkpoint is written only when
          the validation mDice is STRICTLY greater than the best so far (:427-450; the reference's initial best is False == 0).

        `train_batches` / `unlabeled_batches` / `val_batches`: re-iterables (lists, DataLoaders) of (x, y) pairs, the
        unlabelled labels are ignored (:166).  Nothing inside an epoch synchronises with the host; the epoch's scalars are
        fetched with one copy at its end.  Returns the per-epoch records (also handed to `log`)."""
        import itertools
        history = []
        patience = 0
        for epoch in range(int(start_epoch), int(epochs)):
            semi = iter(zip(itertools.cycle(train_batches), itertools.cycle(unlabeled_batches)))
            kept = []
            for _ in range(1, int(iter_per_epoch)):
                (x_l, y_l), u = next(semi)
                x_u = u[0] if isinstance(u, (tuple, list)) else u
                res = self.train_step(x_l.to(self.device), y_l.to(self.device), x_u.to(self.device))
                kept.append(torch.stack([res["loss"].float(), res["sup"].float(), res["unsup"].float()]))
            rec: Dict = {"epoch": epoch, "iter_num": self.iter_num, "lr": float(self.optimizer.param_groups[0]["lr"])}
            if kept:
                sums = torch.stack(kept).sum(0).double().cpu().numpy()
                rec.update(loss=float(sums[0]) / iter_per_epoch, sup=float(sums[1]) / iter_per_epoch,
                           unsup=float(sums[2]) / iter_per_epoch)
                tm = self.epoch_metrics()                       # mean over the epoch's batches
                n = len(kept)
                rec.update(train_miou=tm["miou"] * n / iter_per_epoch, train_mdice=tm["mdice"] * n / iter_per_epoch)
            rec["cw1"], rec["cw2"] = self.consistency_weights()
            val = self.validate(val_batches)
            rec.update({"val_" + k: v for k, v in val.items()})
            self.scheduler.step(val["mdice"])                                                              # :402
            if best_dice < val["mdice"]:                                                                   # :427-433
                best_dice, patience, rec["saved"] = val["mdice"], 0, True
                if checkpoint_path is not None:
                    self.save_checkpoint(checkpoint_path, epoch, best_dice)                                # :440-450
            else:
                patience, rec["saved"] = patience + 1, False
            rec["best_dice"], rec["patience"] = best_dice, patience
            history.append(rec)
            if log is not None:
                log(rec)
        return history

    # -- checkpoint (UAPS_train.py:437-450) --
    def state_for_checkpoint(self, epoch: int, best_dice: float, dataparallel_prefix: bool = True) -> Dict:
        sd = self.model.state_dict()
        if dataparallel_prefix and not any(k.startswith("module.") for k in sd):
            sd = {"module." + k: v for k, v in sd.items()}     # the reference saves nn.DataParallel(model).state_dict()
        # the reference's four keys (UAPS_train.py:443-448) + what a resume needs, all of it plain Python / torch types so that
        # a plain `torch.load(path)` (weights_only=True since torch 2.6) -- the reference's loaders, a user's tools -- reads it
        ck = {"epoch": epoch, "best_dice_1": best_dice, "state_dict": sd, "optimizer": self.optimizer.state_dict(),
              "iter_num": self.iter_num, "scheduler": self.scheduler.state_dict(), "mix_rng": _rng_state_plain(self.mix_rng)}
        if self.step_graph is not None:
            ck["step_key"] = int(self.step_graph.state.key)      # the Philox key of the state-mode perturbation streams
        return ck

    def save_checkpoint(self, path: str, epoch: int, best_dice: float):
        if self.rank == 0:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            torch.save(self.state_for_checkpoint(epoch, best_dice), path)

    def load_checkpoint(self, path: str, load_optimizer: bool = True) -> Dict:
        # map to the CPU: load_state_dict moves the Adam moments to the parameters' device itself, while the per-parameter
        # `step` counters must stay host tensors (optim.Adam reads them without a device sync)
        ck = torch.load(path, map_location="cpu", weights_only=False)
        load_state_dict_any_prefix(self.model, ck["state_dict"])
        if load_optimizer and "optimizer" in ck:
            self.optimizer.load_state_dict(ck["optimizer"])
            for st in self.optimizer.state.values():
                if torch.is_tensor(st.get("step")) and st["step"].device.type != "cpu":
                    st["step"] = st["step"].cpu()
            if "iter_num" in ck:                                   # resume the consistency ramp and the LR plateau state
                self.iter_num = int(ck["iter_num"])
            if "scheduler" in ck:
                self.scheduler.load_state_dict(ck["scheduler"])
            if "mix_rng" in ck:
                self.mix_rng.set_state(_rng_state_numpy(ck["mix_rng"]))
        if self.step_graph is not None:
            # a captured step holds the OLD Adam moment buffers through frozen pointers (load_state_dict replaced them):
            # drop the capture, it is re-recorded after the warm-up steps; the Adam step count is re-read from the loaded state
            self.step_graph.invalidate()
            if "step_key" in ck:
                self.step_graph.state.key = int(ck["step_key"])
        return ck


class BaselineTrainer(UAPSTrainer):
    """The supervised baseline of BASELINE.json configs[0] (baseline/baseline_train.py:100-173): a single-decoder
    U-Net (`net_factory("unet")`), loss = 0.5 * (dice_loss + CrossEntropy) on the labelled batch only, Adam(lr) and
    ReduceLROnPlateau(max, min_lr 1e-7, patience 40) (:102-105).  It is the D = 1 case of the same kernels: no
    perturbations, no mixing, no unlabelled branch."""

    def __init__(self, model: torch.nn.Module, base_lr: float = 1e-3, seed: int = 1337, overlap_comm: bool = True,
                 track_metrics: bool = True):
        super().__init__(model, base_lr=base_lr, seed=seed, overlap_comm=overlap_comm, track_metrics=track_metrics, pair_forward=False)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="max", min_lr=1e-7, patience=40)
        self.n_heads = 1

    def train_step(self, x_l: torch.Tensor, y_l: torch.Tensor, x_u=None, w=None) -> Dict[str, torch.Tensor]:
        if not self.model.training:
            self.model.train()
        with lazybn.scope(), conv.deferred_reduces(model=self.model) as step:
            if self.buckets is not None:
                self.buckets.step = step
            out = self.model(x_l)                                                 # baseline_train.py:158
            main = out[0] if isinstance(out, (tuple, list)) else out
            s = losses.uaps_sup_loss((main,), y_l)                                # :161-164, 0.5 * (dice + CE)
            self.optimizer.zero_grad(set_to_none=True)                            # :166
            s.loss.backward(gradient=self._unit_gradient(s.loss))                 # :168
        if self.buckets is not None:
            self.buckets.finish()
        self.optimizer.step()                                                     # :173
        if self.track_metrics:                                                    # :181-182
            self._cms.append(metrics.seg_confusion(main, y_l))
        self.iter_num += 1
        sc = losses.sup_scalars(s.scalars, 1, main.shape[1])
        self.last = {"loss": s.loss.detach(), "ce": sc["ce"][0], "dice": sc["dice"][0]}
        return self.last


def _rng_state_plain(rng: np.random.RandomState):
    """np.random.RandomState.get_state() with the key array as a torch tensor: nothing a weights-only unpickler refuses."""
    name, keys, pos, has_gauss, cached = rng.get_state()
    return (str(name), torch.from_numpy(np.asarray(keys, dtype=np.uint32).astype(np.int64)), int(pos), int(has_gauss), float(cached))


def _rng_state_numpy(state):
    """Inverse of _rng_state_plain; also accepts the raw numpy tuple older checkpoints of this package hold."""
    name, keys, pos, has_gauss, cached = state
    if torch.is_tensor(keys):
        keys = keys.cpu().numpy()
    return (str(name), np.asarray(keys).astype(np.uint32), int(pos), int(has_gauss), float(cached))


def load_state_dict_any_prefix(model: torch.nn.Module, sd: Dict[str, torch.Tensor]):
    """Accepts the reference's `module.`-prefixed checkpoints (UAPS_model.py:13) and un-prefixed ones."""
    want_prefix = any(k.startswith("module.") for k in model.state_dict())
    has_prefix = any(k.startswith("module.") for k in sd)
    if has_prefix and not want_prefix:
        sd = {k[len("module."):]: v for k, v in sd.items()}
    elif want_prefix and not has_prefix:
        sd = {"module." + k: v for k, v in sd.items()}
    return model.load_state_dict(sd)
