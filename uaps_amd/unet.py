"""Shared-encoder / multi-decoder U-Net of UAPS, parameter-for-parameter compatible with the
reference checkpoint layout (utilities/UAPS_unet.py:31-233; 334 state_dict entries for
UNet_UAPS(3, 4), optionally `module.`-prefixed by the reference's nn.DataParallel wrapper).

Module tree and key names (they ARE the drop-in contract, see tests/test_model_state_dict.py):
  encoder.in_conv.conv_conv.{0,1,4,5}            conv3x3, BN, conv3x3, BN   (UAPS_unet.py:36-44)
  encoder.down{1..4}.maxpool_conv.1.conv_conv.*  max-pool(2) + ConvBlock    (:55-58)
  <decoder>.up{1..4}.conv1x1 / .conv.conv_conv.* conv1x1, bilinear x2 (align_corners), cat, ConvBlock (:72-86)
  <decoder>.out_conv                             conv3x3 -> class logits    (:138-139)
  decoders: main_decoder, aux_decoder1 (FeatureNoise), aux_decoder2 (Dropout), aux_decoder3 (FeatureDropout)

On a ROCm device the convolutions are the MFMA implicit-GEMM kernels of csrc/conv_kernels.hpp and everything between them --
BatchNorm(train)+LeakyReLU+Dropout, bilinear-x2+concat, the three feature perturbations -- and
everything after the logits are the HIP kernels of this package (csrc/*.hip).
"""
from __future__ import annotations

import os
import weakref
from typing import List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import bounds, config, conv, fused, perturb, stepctx

FEATURE_CHANNELS = (16, 32, 64, 128, 256)          # UAPS_unet.py:213
ENCODER_DROPOUT = (0.05, 0.1, 0.2, 0.3, 0.5)       # UAPS_unet.py:214
LEAKY_SLOPE = 0.01                                  # nn.LeakyReLU() default
_EPILOGUE_STATS = config.flag("UAPS_EPILOGUE_STATS", True)   # A/B switches for tools/ab_bench.sh
_VIRTUAL_CAT = config.flag("UAPS_VIRTUAL_CAT", True)
_FUSED_FAN = config.flag("UAPS_FUSED_FAN", True)
_DECODER_CHAINS = max(2, config.integer("UAPS_DECODER_CHAINS", 8))      # diagnosis: decoders are dealt round-robin onto this many streams (main included)
_FAN_BESIDE = config.flag("UAPS_FAN_BESIDE", True)        # the perturbed feature copies are written on a side stream beside the encoder's next levels
_PACK_BESIDE = config.flag("UAPS_PACK_BESIDE", True)      # the decoders' weights are packed on a side stream beside the encoder's forward
_FUSED_POOL = config.flag("UAPS_FUSED_POOL", True)
_FUSED_BN_CONV = config.flag("UAPS_FUSED_BN_CONV", True)
# One HIP stream per auxiliary decoder (see UNet_UAPS.forward): +5 % images/s on the bench step, opt-in because kernels of
# different decoders then overlap and per-launch timings (bench.py's roofline, rocprof averages) stop describing one kernel.
_DECODER_STREAMS = config.flag("UAPS_DECODER_STREAMS", False)
_CAPTURE_KEEP = weakref.WeakKeyDictionary()     # model -> tensors that cross streams inside its captured step (UNet_UAPS.forward)
# BatchNorm partial sums in the conv epilogues are taken about running_mean - conv bias (no variance cancellation for channels
# with |mean| >> std).  Off: plain sums, as in round 1 -- then forward_pair and two successive forwards agree bit for bit (with
# the shift the second forward already sees the running mean the first one updated: another rounding of the same statistics)
_STAT_SHIFT = config.flag("UAPS_STAT_SHIFT", True)


class ConvBlock(nn.Module):
    """conv3x3 + BN + LeakyReLU + Dropout(p) + conv3x3 + BN + LeakyReLU (UAPS_unet.py:31-47)."""

    def __init__(self, in_channels: int, out_channels: int, dropout_p: float):
        super().__init__()
        layers = [nn.Conv2d(in_channels, out_channels, 3, padding=1), nn.BatchNorm2d(out_channels),
                  nn.LeakyReLU(LEAKY_SLOPE), nn.Dropout(dropout_p),
                  nn.Conv2d(out_channels, out_channels, 3, padding=1), nn.BatchNorm2d(out_channels),
                  nn.LeakyReLU(LEAKY_SLOPE)]
        self.conv_conv = nn.Sequential(*layers)      # indices 0,1,4,5 carry the parameters

    def forward(self, x: torch.Tensor, x2: Optional[torch.Tensor] = None, lazy: bool = False, x2_low: bool = False):
        """x2: second half of a channel concatenation [x, x2] that is not materialised (GPU only).
        x2_low: x2 is the LOW-resolution tensor; the first convolution's kernels up-sample it x2 while staging (conv.conv2d_cat up2).
        lazy: the caller feeds the result to exactly one convolution and can apply the last BatchNorm + LeakyReLU there
        (fused.bn_act_conv); then a PendingBnAct may come back instead of a tensor (train mode on the GPU only)."""
        if not x.is_cuda:
            return self.conv_conv(x if x2 is None else torch.cat([x, x2], dim=1))   # plain torch modules (CPU tests)
        # GPU: MFMA implicit-GEMM convs (csrc/conv_kernels.hpp) without bias (train-mode BN cancels it; the
        # fused kernel folds it into running_mean / the eval shift) + the fused BN+LeakyReLU+Dropout kernels
        c0, b0, _, d0, c1, b1, _ = self.conv_conv
        if self.training and _EPILOGUE_STATS:      # batch statistics: their first pass rides in the conv epilogue
            # the epilogue sums are taken about running_mean - conv bias of the BatchNorm they feed (no variance cancellation)
            sh0, sh1 = ((b0.running_mean, c0.bias), (b1.running_mean, c1.bias)) if _STAT_SHIFT else (None, None)
            y, st = (conv.conv2d_with_stats(x, c0.weight, None, stat_shift=sh0) if x2 is None
                     else conv.conv2d_cat(x, x2, c0.weight, None, True, stat_shift=sh0, up2=x2_low))
            if d0.p == 0.0 and _FUSED_BN_CONV and fused.can_fuse_bn_into_conv(y, c1.weight):
                # no dropout in between (decoder blocks): the second conv normalises + activates while staging its input
                y, st = fused.bn_act_conv(y, st, c0.bias, b0, LEAKY_SLOPE, c1.weight, None, want_stats=True, stat_shift=sh1)
                if lazy and y.shape[1] > 4:
                    return PendingBnAct(y, st, c1.bias, b1)
                return fused.bn_act(y, c1.bias, b1, LEAKY_SLOPE, 0.0, True, st)
            a = fused.bn_act(y, c0.bias, b0, LEAKY_SLOPE, d0.p, True, st)
            y, st = conv.conv2d_with_stats(a, c1.weight, None, stat_shift=sh1)
            return fused.bn_act(y, c1.bias, b1, LEAKY_SLOPE, 0.0, True, st)
        y = conv.conv2d(x, c0.weight, None) if x2 is None else conv.conv2d_cat(x, x2, c0.weight, None, up2=x2_low)
        a = fused.bn_act(y, c0.bias, b0, LEAKY_SLOPE, d0.p, self.training)
        return fused.bn_act(conv.conv2d(a, c1.weight, None), c1.bias, b1, LEAKY_SLOPE, 0.0, self.training)


class PendingBnAct:
    """Raw output of a ConvBlock's second conv with its epilogue statistics: leaky_relu(bn(y + bias)) still to be applied,
    by the one convolution that consumes it (`conv`) or explicitly (`materialize`)."""

    def __init__(self, y, stats, conv_bias, bn):
        self.y, self.stats, self.conv_bias, self.bn = y, stats, conv_bias, bn

    def conv(self, weight, bias):
        if fused.can_fuse_bn_into_conv(self.y, weight):
            return fused.bn_act_conv(self.y, self.stats, self.conv_bias, self.bn, LEAKY_SLOPE, weight, bias)
        return conv.conv2d(self.materialize(), weight, bias)

    def materialize(self):
        return fused.bn_act(self.y, self.conv_bias, self.bn, LEAKY_SLOPE, 0.0, True, self.stats)


class DownBlock(nn.Module):
    """max-pool 2x2 then ConvBlock (UAPS_unet.py:50-62)."""

    def __init__(self, in_channels: int, out_channels: int, dropout_p: float):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), ConvBlock(in_channels, out_channels, dropout_p))

    def forward(self, x):
        return self.maxpool_conv(x)


class UpBlock(nn.Module):
    """conv1x1 on the coarse map, bilinear x2 (align_corners=True), concat [skip, up], ConvBlock.

    The reference's Decoder never forwards its `bilinear` flag (UAPS_unet.py:129-136), so the
    bilinear branch (72-75) is the one every shipped checkpoint has; that is what is built here."""

    def __init__(self, in_channels1: int, in_channels2: int, out_channels: int, dropout_p: float = 0.0):
        super().__init__()
        self.conv1x1 = nn.Conv2d(in_channels1, in_channels2, kernel_size=1)
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.conv = ConvBlock(in_channels2 * 2, out_channels, dropout_p)

    def forward(self, coarse, skip, lazy: bool = False):
        """coarse: tensor or PendingBnAct (the previous UpBlock's output before its last BatchNorm + LeakyReLU);
        lazy: the caller accepts a PendingBnAct (see ConvBlock.forward)."""
        if not isinstance(coarse, PendingBnAct) and not coarse.is_cuda:
            return self.conv(torch.cat([skip, self.up(self.conv1x1(coarse))], dim=1))
        # up4 at the metric's size: the up-sampled tensor is never written -- the ConvBlock's first convolution reads the 1x1
        # projection's LOW-resolution output and up-samples it while staging (conv.conv2d_cat up2); that operand's bound is the
        # projection's own max|output| (it has no BatchNorm behind it), tracked in its epilogue on request
        fuse_up = _VIRTUAL_CAT and skip.shape[1] % 16 == 0 and conv.up2_eligible(skip, self.conv.conv_conv[0].weight)
        if fuse_up:
            conv.request_out_amax()
        if isinstance(coarse, PendingBnAct):
            low = coarse.conv(self.conv1x1.weight, self.conv1x1.bias)
        else:
            low = conv.conv2d(coarse, self.conv1x1.weight, self.conv1x1.bias)
        if fuse_up:
            am = conv.take_out_amax()
            if am is not None and low.shape[2] * 2 == skip.shape[2] and low.shape[3] * 2 == skip.shape[3]:
                return self.conv(skip, bounds.put(low, am), lazy=lazy, x2_low=True)
        if _VIRTUAL_CAT and skip.shape[1] % 16 == 0:
            return self.conv(skip, fused.upsample2x(low), lazy=lazy)     # the conv kernels read [skip | up] as two tensors
        return self.conv(fused.up_cat(skip, low), lazy=lazy)     # bilinear x2 written straight into the concat buffer


class Encoder(nn.Module):
    """Five-scale feature pyramid (UAPS_unet.py:89-116)."""

    def __init__(self, in_chns: int, feature_chns: Sequence[int] = FEATURE_CHANNELS,
                 dropout: Sequence[float] = ENCODER_DROPOUT):
        super().__init__()
        if len(feature_chns) != 5:
            raise AssertionError("five feature scales expected")       # UAPS_unet.py:98
        f = list(feature_chns)
        self.in_conv = ConvBlock(in_chns, f[0], dropout[0])
        self.down1 = DownBlock(f[0], f[1], dropout[1])
        self.down2 = DownBlock(f[1], f[2], dropout[2])
        self.down3 = DownBlock(f[2], f[3], dropout[3])
        self.down4 = DownBlock(f[3], f[4], dropout[4])

    def forward(self, x) -> List[torch.Tensor]:
        feats = [self.in_conv(x)]
        for blk in (self.down1, self.down2, self.down3, self.down4):
            feats.append(blk(feats[-1]))
        return feats


class Decoder(nn.Module):
    """Four UpBlocks + 3x3 classifier (UAPS_unet.py:119-153)."""

    def __init__(self, class_num: int, feature_chns: Sequence[int] = FEATURE_CHANNELS):
        super().__init__()
        if len(feature_chns) != 5:
            raise AssertionError("five feature scales expected")       # UAPS_unet.py:127
        f = list(feature_chns)
        self.up1 = UpBlock(f[4], f[3], f[3])
        self.up2 = UpBlock(f[3], f[2], f[2])
        self.up3 = UpBlock(f[2], f[1], f[1])
        self.up4 = UpBlock(f[1], f[0], f[0])
        self.out_conv = nn.Conv2d(f[0], class_num, kernel_size=3, padding=1)

    def forward(self, feats: Sequence[torch.Tensor]) -> torch.Tensor:
        lazy = _FUSED_BN_CONV       # each UpBlock output feeds exactly one conv: the next conv1x1 or out_conv
        x = self.up1(feats[4], feats[3], lazy)
        x = self.up2(x, feats[2], lazy)
        x = self.up3(x, feats[1], lazy)
        x = self.up4(x, feats[0], lazy)
        if isinstance(x, PendingBnAct):
            return x.conv(self.out_conv.weight, self.out_conv.bias)
        if not x.is_cuda:
            return self.out_conv(x)
        return conv.conv2d(x, self.out_conv.weight, self.out_conv.bias)


class UNet(nn.Module):
    """Single-decoder U-Net (UAPS_unet.py:188-205; same blocks as utilities/baseline_unet.py:159-176)."""

    def __init__(self, in_chns: int, class_num: int, feature_chns: Sequence[int] = FEATURE_CHANNELS):
        super().__init__()
        self.encoder = Encoder(in_chns, feature_chns)
        self.decoder = Decoder(class_num, feature_chns)
        self._bns = None

    def forward(self, x):
        if x.is_cuda and self.training:          # bounds of the train-mode BatchNorm outputs (conv mode 'h16')
            if self._bns is None:
                self._bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
            bounds.refresh(self._bns)
        return self.decoder(self.encoder(x))


# perturbation applied in front of auxiliary decoder i (UAPS_unet.py:227-231); decoders beyond the
# reference's three (the K=5 stress config has no reference implementation) cycle through the same three.
_PERTURBATIONS = ("noise", "dropout", "feature_dropout")


class UNet_UAPS(nn.Module):
    """Shared encoder, main decoder on clean features, `n_aux` auxiliary decoders on perturbed
    features; returns (main, aux1, ..., aux_n) logits, each [B, class_num, H, W] fp32
    (UAPS_unet.py:208-233).  `n_aux=3` is the reference model."""

    def __init__(self, in_chns: int, class_num: int, n_aux: int = 3,
                 feature_chns: Sequence[int] = FEATURE_CHANNELS, dropout: Sequence[float] = ENCODER_DROPOUT):
        super().__init__()
        if not 0 <= n_aux <= 7:
            raise ValueError("n_aux must be in 0..7")
        self.n_aux = n_aux
        self.encoder = Encoder(in_chns, feature_chns, dropout)
        self.main_decoder = Decoder(class_num, feature_chns)
        for i in range(1, n_aux + 1):
            setattr(self, f"aux_decoder{i}", Decoder(class_num, feature_chns))
        self._noise = perturb.FeatureNoise()
        self._conv_weights = None
        self._pack_split = None
        self._fan_stream = None
        self._bns = None
        self._streams = None

    def decoder_parameters(self):
        """The parameters whose gradients are complete when the backward of the deepest feature's fan-out runs (every decoder's whole
        backward is in front of it): UAPSTrainer steps them beside the encoder's backward."""
        return [p for dec in [self.main_decoder] + self.aux_decoders() for p in dec.parameters()]

    def aux_decoders(self) -> List[Decoder]:
        return [getattr(self, f"aux_decoder{i}") for i in range(1, self.n_aux + 1)]

    def _perturb(self, kind: str, feats, groups: int = 1):
        if kind == "noise":
            return [self._noise(f, groups=groups) for f in feats]
        if kind == "dropout":
            return [perturb.Dropout(f) for f in feats]
        return [perturb.FeatureDropout(f, groups=groups) for f in feats]

    def forward_main(self, x):
        """Main-head logits only: the shared encoder and the main decoder on clean features (UAPS_unet.py:224-226), the auxiliary
        decoders not run -- the evaluation path's cost when only the mask is wanted (notebook cells 11-13 use `output` alone)."""
        if x.is_cuda:
            if self._conv_weights is None:
                self._conv_weights = [m.weight for m in self.modules() if isinstance(m, nn.Conv2d)]
            conv.pack_all(self._conv_weights)
            if self.training:
                if self._bns is None:
                    self._bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
                bounds.refresh(self._bns)
        return self.main_decoder(self.encoder(x))

    def forward_pair(self, x_a, x_b, perturbations=None):
        """The two forwards of a training step (UAPS_train.py:177 labelled, :185 unlabelled) as ONE pass over the
        concatenated batch: every convolution runs once on 2B images, while everything the reference computes per
        forward call keeps its per-call meaning -- train-mode BatchNorm statistics (and the two successive
        running-statistics updates), the FeatureNoise tensor and the FeatureDropout threshold are per half.
        Returns D logit tensors of shape [2B, class_num, H, W]; rows [:B] belong to x_a, rows [B:] to x_b.
        GPU only (the grouped kernels have no PyTorch counterpart)."""
        if x_a.shape != x_b.shape:
            raise ValueError("forward_pair: the two batches must have the same shape")
        with fused.stat_groups(2):
            return self.forward(fused.cat_batches(x_a, x_b), perturbations, _groups=2)

    def forward(self, x, perturbations=None, _groups: int = 1):
        """`perturbations`: optional list (one entry per auxiliary decoder) of callables
        feats -> feats replacing the random draws (parity tests inject recorded draws here)."""
        packing = None
        if x.is_cuda:                            # all conv weights packed by one launch (once per optimizer step)
            if self._conv_weights is None:
                self._conv_weights = [m.weight for m in self.modules() if isinstance(m, nn.Conv2d)]
            if _PACK_BESIDE and _DECODER_STREAMS and perturbations is None and self.n_aux > 0 and _FUSED_FAN:
                # the encoder's weights now, the decoders' (70 % of the packing) on a side stream beside the encoder's forward
                if self._pack_split is None:
                    enc = {id(m.weight) for m in self.encoder.modules() if isinstance(m, nn.Conv2d)}
                    self._pack_split = ([w for w in self._conv_weights if id(w) in enc], [w for w in self._conv_weights if id(w) not in enc])
                conv.pack_all(self._pack_split[0])
                packing = conv.pack_all_beside(self._pack_split[1], x.device)
            else:
                conv.pack_all(self._conv_weights)
            if self.training:                        # bounds of the train-mode BatchNorm outputs (conv mode 'h16')
                if self._bns is None:
                    self._bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
                bounds.refresh(self._bns)
        if x.is_cuda and perturbations is None and self.n_aux > 0 and _FUSED_FAN:
            # per feature map: the clean handle + one perturbed copy per auxiliary decoder (+ the 2x2 max-pool that feeds
            # the next encoder level); all their gradients come back through ONE kernel that re-applies the perturbations,
            # routes the pooled gradient to its arg-max positions and sums (perturb._PerturbFan)
            kinds = [_PERTURBATIONS[i % 3] for i in range(self.n_aux)]
            enc = self.encoder
            fan_side = None
            if _DECODER_STREAMS and _FAN_BESIDE:
                # the perturbed copies of a feature map feed the auxiliary decoders only: written on a side stream beside the encoder's
                # next levels (the max-pool, which the next level waits for, is launched first on this stream)
                if self._fan_stream is None or self._fan_stream.device != x.device:
                    self._fan_stream = torch.cuda.Stream(device=x.device)
                fan_side = self._fan_stream
            fans, f = [], enc.in_conv(x)
            stepctx.fwd().fan_side = fan_side
            try:
                for blk in (enc.down1, enc.down2, enc.down3, enc.down4, None):
                    # (the fan-in kernel sums at most 8 gradients: clean + n_aux perturbed + the pooled one)
                    pool = blk is not None and _FUSED_POOL and f.shape[2] % 2 == 0 and f.shape[3] % 8 == 0 and self.n_aux + 2 <= 8
                    fan = perturb.perturbed_fan_out(f, kinds, _groups, self._noise.uniform_range, with_pool=pool)
                    fans.append(fan)
                    if blk is not None:
                        f = blk.maxpool_conv[1](fan[-1]) if pool else blk(f)
            finally:
                stepctx.fwd().fan_side = None
            if fan_side is not None:
                torch.cuda.current_stream(x.device).wait_stream(fan_side)
            per_dec = [[fan[d] for fan in fans] for d in range(1 + self.n_aux)]
            decoders = [self.main_decoder] + self.aux_decoders()
            if packing is not None:
                torch.cuda.current_stream(x.device).wait_stream(packing)      # the decoders' packed weights
            if not _DECODER_STREAMS:
                return tuple(dec(per_dec[d]) for d, dec in enumerate(decoders))
            # The decoders are independent chains of ~60 launches each: every auxiliary decoder gets its own HIP stream
            # (forward here, and autograd runs each backward node on its forward stream), so one decoder's small
            # dependent kernels (BatchNorm finalize, weight-gradient reduce) and the tail of its convolutions overlap the
            # other decoders' convolutions instead of leaving most CUs idle.
            main = torch.cuda.current_stream(x.device)
            if self._streams is None or len(self._streams) != self.n_aux or self._streams[0].device != x.device:
                self._streams = [torch.cuda.Stream(device=x.device) for _ in range(self.n_aux)]
            ready = main.record_event()
            outs = [None] * len(decoders)
            # Memory that crosses streams.  The caching allocator hands a freed block back to the stream it was allocated on at
            # once: the perturbed feature maps (allocated on the main stream) are read by the side stream's kernels -- in the
            # forward, and as saved convolution inputs in the backward, where autograd drops them as soon as the node's kernels
            # are ENQUEUED -- so without a record the main decoder's next allocation could reuse, and overwrite, a map a side
            # stream's weight-gradient kernel has yet to read (seen as a rare bit mismatch between two runs of the same steps).
            # Eager: record_stream (the block is reusable only after the other stream has passed the point of the free).
            # Under capture the same reuse would be baked into the graph with no dependency between the two kernels: the tensors
            # are simply kept until the next capture (their private-pool addresses are the graph's own anyway).
            capturing = torch.cuda.is_current_stream_capturing()
            keep = []
            on_main = []
            for d in range(1, len(decoders)):
                if d % _DECODER_CHAINS == 0:         # (diagnosis: fewer chains than decoders -- this one shares the main stream)
                    on_main.append(d)
                    continue
                side = self._streams[d % _DECODER_CHAINS - 1]
                side.wait_event(ready)
                for t in per_dec[d]:
                    if capturing:
                        keep.append(t.detach())      # the STORAGE is what must stay; a tensor with its grad_fn would keep the captured step's
                    else:                            # whole autograd graph alive (and with it the parameters' AccumulateGrad nodes, created on
                        t.record_stream(side)        # the capture's streams: PyTorch warns when a later eager step meets them on another stream)
                with torch.cuda.stream(side):
                    outs[d] = decoders[d](per_dec[d])
                if capturing:
                    keep.append(outs[d].detach())
                else:
                    outs[d].record_stream(main)          # consumed by the loss on the main stream
            if capturing:
                _CAPTURE_KEEP[self] = keep              # not an attribute: the model stays deep-copyable
            outs[0] = decoders[0](per_dec[0])
            for d in on_main:
                outs[d] = decoders[d](per_dec[d])
            for side in self._streams:
                main.wait_stream(side)
            return tuple(outs)
        feats = self.encoder(x)
        if x.is_cuda and torch.is_grad_enabled() and self.n_aux > 0:
            # every decoder gets its own handle on each feature map: the 1 + n_aux gradients are summed by one kernel
            handles = [fused.fan_out(f, 1 + self.n_aux) if f.requires_grad else (f,) * (1 + self.n_aux) for f in feats]
            per_dec = [[h[d] for h in handles] for d in range(1 + self.n_aux)]
        else:
            per_dec = [feats] * (1 + self.n_aux)
        outs = [self.main_decoder(per_dec[0])]
        for i, dec in enumerate(self.aux_decoders()):
            if perturbations is not None and perturbations[i] is not None:
                pf = perturbations[i](per_dec[i + 1])
            else:
                pf = self._perturb(_PERTURBATIONS[i % 3], per_dec[i + 1], _groups)
            outs.append(dec(pf))
        return tuple(outs)
