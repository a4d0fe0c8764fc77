"""ResNet backbones of the reference (utilities/resnet.py:98-183, 192-213) and a ResNet-encoder UAPS network
(BASELINE.json configs[4]).

`ResNet` has the reference's module tree and therefore its state_dict keys (`conv1`, `bn1`, `layerN.M.convK / bnK`,
`layerN.0.downsample.0 / .1`), including the reference's choice of replacing the stride of layer3/layer4 by dilation
for the Bottleneck nets (resnet.py:201-203: c1..c4 = 256/512/1024/2048 channels at strides 4/8/8/8).  On a ROCm
device every stride-1 convolution (all 1x1 projections, all dilated and plain 3x3) runs on the MFMA kernels of
csrc/conv_kernels.hpp with the BatchNorm statistics in their epilogue, BatchNorm + ReLU are the fused kernels of
csrc/norm_act.hip (LeakyReLU slope 0 = ReLU, slope 1 = identity) and the residual joins one `relu(a + b)` kernel whose
backward also sums the gradients of the join's consumers (add_relu below).  Of the three strided convolutions, layer2's 3x3
runs as a stride-1 3x3 over the four sampling phases of its input (conv.conv3x3s2) and its 1x1 shortcut as a stride-1
projection of every second row and column (conv.subsample2), both on the MFMA kernels; the 7x7 stem and the stem's 3x3
max-pool run on the kernels of csrc/conv_strided.hip.  No library convolution, pooling or elementwise kernel is left on this
path.

The reference has no UAPS model on this backbone (`utilities/base.py` is abstract), so the
decoder below is this build's design: parity is pinned for the backbone only (tests/golden/g7_resnet.npz).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, bounds, conv, fused, perturb
from .unet import ConvBlock, UpBlock, _PERTURBATIONS


# ---- relu(a + b) with the hand-written kernels ---------------------------------------------------------------------

class _AddRelu(torch.autograd.Function):
    """relu(a + b) handed out as `n` handles on ONE tensor (the join's consumers: the next block's first convolution, its shortcut,
    a decoder); the backward sums the n incoming gradients and masks them in one pass (uaps_relu_bwd_sum) -- without the handles
    the autograd engine adds the gradients with a pass of its own (an ATen kernel) in front of the mask."""

    @staticmethod
    def forward(ctx, a, b, am, n):
        _lib.require_device(a, "add_relu")
        ctx.set_materialize_grads(False)
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        with _lib.device_guard(a.device):
            rc = _lib.lib().uaps_add_relu_h(_lib.mk_hints((), am) if am is not None else None, a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _lib.current_stream(a.device))
        _lib.check(rc, "uaps_add_relu")
        ctx.save_for_backward(out)
        return tuple(out.view_as(out) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        (out,) = ctx.saved_tensors
        gs = [g.contiguous() for g in grads if g is not None]
        if not gs:
            return None, None, None, None
        dx = torch.empty_like(gs[0])
        ptrs = (C.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
        with _lib.device_guard(dx.device):
            rc = _lib.lib().uaps_relu_bwd_sum(ptrs, len(gs), out.data_ptr(), dx.data_ptr(), dx.numel(), _lib.current_stream(dx.device))
        _lib.check(rc, "uaps_relu_bwd_sum")
        return dx, dx, None, None


def add_relu(a: torch.Tensor, b: torch.Tensor, n: int = 1):
    """relu(a + b); n > 1: a tuple of n handles on the result, one per consumer (see _AddRelu)."""
    if not a.is_cuda:
        out = F.relu(a + b)
        return out if n == 1 else (out,) * n
    if not 1 <= n <= 4:
        raise ValueError("add_relu: 1..4 consumers")
    am = bounds.new_amax(a.device) if bounds.enabled() else None      # max(out), raised by the kernel: the join feeds the next block's convolutions
    outs = tuple(bounds.put(o, am) for o in _AddRelu.apply(a, b, am, n))
    return outs[0] if n == 1 else outs


def conv_bn_act(x: torch.Tensor, cv: nn.Conv2d, bn: nn.BatchNorm2d, relu: bool, training: bool) -> torch.Tensor:
    """[relu](bn(conv(x))) for a bias-free convolution.  GPU: the tiled MFMA kernels for every stride-1 1x1 / 3x3 (dilation 1, 2,
    4) convolution, the general strided kernels for the rest; BatchNorm(+ReLU) always the fused HIP kernels."""
    if not x.is_cuda:
        y = bn(cv(x))
        return F.relu(y) if relu else y
    slope = 0.0 if relu else 1.0
    k, d = cv.kernel_size[0], cv.dilation[0]
    if k == 1 and cv.stride == (2, 2) and cv.padding == (0, 0) and d == 1 and cv.groups == 1 and cv.bias is None:
        # the strided shortcut projection (resnet.py:157-161) = a stride-1 projection of every second row and column
        x = conv.subsample2(x)
        if training:
            y, st = conv.conv2d_with_stats(x, cv.weight, None)
            return fused.bn_act(y, None, bn, slope, 0.0, True, st)
        return fused.bn_act(conv.conv2d(x, cv.weight, None), None, bn, slope, 0.0, False)
    if k == 3 and cv.stride == (2, 2) and cv.padding == (1, 1) and d == 1 and cv.groups == 1 and cv.bias is None and conv.conv3x3s2_supported(x):
        # the strided 3x3 of layer2.0 (resnet.py:147 via 8-10): stride 1 over the four sampling phases
        if training:
            y, st = conv.conv3x3s2(x, cv.weight, with_stats=True)
            return fused.bn_act(y, None, bn, slope, 0.0, True, st)
        return fused.bn_act(conv.conv3x3s2(x, cv.weight), None, bn, slope, 0.0, False)
    own = cv.stride == (1, 1) and cv.groups == 1 and ((k == 1 and d == 1) or (k == 3 and d in (1, 2, 4) and cv.padding == (d, d))) \
        and not (k == 3 and d > 1 and cv.in_channels <= 4)
    if own and training:
        y, st = conv.conv2d_with_stats(x, cv.weight, None, dilation=d)
        return fused.bn_act(y, None, bn, slope, 0.0, True, st)
    if own:
        y = conv.conv2d(x, cv.weight, None, dilation=d)
    else:             # the strided layers (7x7 / 2 stem, 3x3 / 2, 1x1 / 2 shortcut) and odd shapes: the general kernels of csrc/conv_strided.hip
        if cv.groups != 1 or d != 1 or cv.stride[0] != cv.stride[1] or cv.padding[0] != cv.padding[1] or cv.bias is not None:
            raise ValueError("conv_bn_act: grouped / dilated-strided / asymmetric convolutions are not part of the reference's ResNet")
        y = conv.conv2d_strided(x, cv.weight, cv.stride[0], cv.padding[0])
    return fused.bn_act(y, None, bn, slope, 0.0, training)


def conv_bn_add_relu(x: torch.Tensor, cv: nn.Conv2d, bn: nn.BatchNorm2d, identity: torch.Tensor, n: int, training: bool):
    """relu(bn(conv(x)) + identity), the end of a residual block (resnet.py:44-50, 85-91).  Training on the GPU with a stride-1
    convolution of the tiled kernels: the join happens in the BatchNorm's apply pass (fused.bn_add_relu) -- the normalised tensor is
    never written; otherwise the separate kernels."""
    k, d = cv.kernel_size[0], cv.dilation[0]
    own = cv.stride == (1, 1) and cv.groups == 1 and cv.bias is None and ((k == 1 and d == 1) or (k == 3 and d in (1, 2, 4) and cv.padding == (d, d))) \
        and not (k == 3 and d > 1 and cv.in_channels <= 4)
    if x.is_cuda and training and own and torch.is_grad_enabled():
        y, st = conv.conv2d_with_stats(x, cv.weight, None, dilation=d)
        return fused.bn_add_relu(y, st, bn, identity, n)
    return add_relu(conv_bn_act(x, cv, bn, False, training), identity, n)


def _conv3x3(inp, out, stride=1, dilation=1):
    return nn.Conv2d(inp, out, 3, stride=stride, padding=dilation, bias=False, dilation=dilation)      # resnet.py:8-10


def _conv1x1(inp, out, stride=1):
    return nn.Conv2d(inp, out, 1, stride=stride, bias=False)                                           # resnet.py:13-14


class BasicBlock(nn.Module):
    """utilities/resnet.py:17-52."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride, dilation)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, x_id=None, n_out: int = 1):
        """x_id: a second handle on the input for the shortcut, n_out: handles wanted on the output (add_relu)."""
        x_id = x if x_id is None else x_id
        out = conv_bn_act(x, self.conv1, self.bn1, True, self.training)
        identity = x_id if self.downsample is None else conv_bn_act(x_id, self.downsample[0], self.downsample[1], False, self.training)
        return conv_bn_add_relu(out, self.conv2, self.bn2, identity, n_out, self.training)


class Bottleneck(nn.Module):
    """utilities/resnet.py:55-95: 1x1 reduce, 3x3 (stride / dilation), 1x1 expand x4, shortcut, ReLU."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = _conv1x1(inplanes, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv3x3(planes, planes, stride, dilation)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv1x1(planes, planes * self.expansion)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, x_id=None, n_out: int = 1):
        """x_id: a second handle on the input for the shortcut, n_out: handles wanted on the output (add_relu)."""
        x_id = x if x_id is None else x_id
        out = conv_bn_act(x, self.conv1, self.bn1, True, self.training)
        out = conv_bn_act(out, self.conv2, self.bn2, True, self.training)
        identity = x_id if self.downsample is None else conv_bn_act(x_id, self.downsample[0], self.downsample[1], False, self.training)
        return conv_bn_add_relu(out, self.conv3, self.bn3, identity, n_out, self.training)


class ResNet(nn.Module):
    """utilities/resnet.py:98-183 without the classifier (the reference's `base_forward` never uses one)."""

    def __init__(self, block, layers: Sequence[int], replace_stride_with_dilation: Optional[Sequence[bool]] = None, in_chns: int = 3,
                 zero_init_residual: bool = False):
        super().__init__()
        self.channels = [64 * block.expansion, 128 * block.expansion, 256 * block.expansion, 512 * block.expansion]
        rsd = list(replace_stride_with_dilation) if replace_stride_with_dilation is not None else [False, False, False]
        if len(rsd) != 3:
            raise ValueError("replace_stride_with_dilation should be None or a 3-element tuple, got {}".format(rsd))
        self.inplanes, self.dilation = 64, 1
        self.conv1 = nn.Conv2d(in_chns, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, dilate=rsd[0])
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2, dilate=rsd[1])
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2, dilate=rsd[2])
        for m in self.modules():                                      # resnet.py:136-141
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)
                elif isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        previous_dilation, downsample = self.dilation, None
        if dilate:                                                    # resnet.py:154-156: the stride becomes a dilation
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(_conv1x1(self.inplanes, planes * block.expansion, stride), nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, previous_dilation)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes, dilation=self.dilation) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def base_forward(self, x):
        """(c1, c2, c3, c4), resnet.py:171-182."""
        x = conv_bn_act(x, self.conv1, self.bn1, True, self.training)
        x = conv.maxpool3x3s2(x) if x.is_cuda else self.maxpool(x)
        if not (x.is_cuda and torch.is_grad_enabled() and x.requires_grad):
            c1 = self.layer1(x)
            c2 = self.layer2(c1)
            c3 = self.layer3(c2)
            c4 = self.layer4(c3)
            return c1, c2, c3, c4
        # training on the GPU: every residual join hands one handle to each of its consumers (next block's first convolution, its
        # shortcut, and -- for a layer's last block -- the feature tuple), so their gradients meet inside the join's backward pass
        feats = []
        h = fused.fan_out(x, 2)                           # the pooled stem output feeds layer1.0's convolution and its shortcut
        layers = (self.layer1, self.layer2, self.layer3, self.layer4)
        for li, layer in enumerate(layers):
            for bi, blk in enumerate(layer):
                last = bi == len(layer) - 1
                n_out = (1 if li == len(layers) - 1 else 3) if last else 2
                h = blk(h[0], h[1], n_out)
                h = h if isinstance(h, tuple) else (h,)
            feats.append(h[-1])
        return tuple(feats)

    forward = base_forward


def resnet18(in_chns: int = 3):
    return ResNet(BasicBlock, [2, 2, 2, 2], in_chns=in_chns)


def resnet34(in_chns: int = 3):
    return ResNet(BasicBlock, [3, 4, 6, 3], in_chns=in_chns)


def resnet50(in_chns: int = 3):
    return ResNet(Bottleneck, [3, 4, 6, 3], [False, True, True], in_chns=in_chns)       # resnet.py:200-203


def resnet101(in_chns: int = 3):
    return ResNet(Bottleneck, [3, 4, 23, 3], [False, True, True], in_chns=in_chns)


def resnet152(in_chns: int = 3):
    return ResNet(Bottleneck, [3, 8, 36, 3], [False, True, True], in_chns=in_chns)


# ---- UAPS on a dilated Bottleneck ResNet ------------------------------------------------------------------------------

class _Proj(nn.Module):
    """1x1 projection + BatchNorm + LeakyReLU of one backbone feature map."""

    def __init__(self, inp, out):
        super().__init__()
        self.conv = nn.Conv2d(inp, out, 1, bias=False)
        self.bn = nn.BatchNorm2d(out)

    def forward(self, x):
        if not x.is_cuda:
            return F.leaky_relu(self.bn(self.conv(x)), 0.01)
        if self.training:
            y, st = conv.conv2d_with_stats(x, self.conv.weight, None)
            return fused.bn_act(y, None, self.bn, 0.01, 0.0, True, st)
        return fused.bn_act(conv.conv2d(x, self.conv.weight, None), None, self.bn, 0.01, 0.0, False)


class ResDecoder(nn.Module):
    """Decoder for (c1 @ 1/4, c2, c3, c4 @ 1/8): the three 1/8 maps are projected to `mid` channels and merged by two
    ConvBlocks over never-materialised concatenations, the result is up-sampled x2 and merged with the projected c1 by an
    UpBlock (the reference's decoder step, UAPS_unet.py:65-86), a 3x3 classifier follows at 1/4 resolution and two
    bilinear x2 up-samplings (align_corners=True) bring the logits to the input resolution."""

    def __init__(self, channels: Sequence[int], class_num: int, mid: int = 128, low: int = 64):
        super().__init__()
        self.p1, self.p2, self.p3, self.p4 = _Proj(channels[0], low), _Proj(channels[1], mid), _Proj(channels[2], mid), _Proj(channels[3], mid)
        self.merge34 = ConvBlock(2 * mid, mid, 0.0)
        self.merge234 = ConvBlock(2 * mid, mid, 0.0)
        self.up = UpBlock(mid, low, low)
        self.out_conv = nn.Conv2d(low, class_num, kernel_size=3, padding=1)

    def forward(self, feats):
        c1, c2, c3, c4 = feats
        a = self.merge34(self.p3(c3), self.p4(c4))
        b = self.merge234(self.p2(c2), a)
        x = self.up(b, self.p1(c1))
        if not x.is_cuda:
            z = self.out_conv(x)
            return F.interpolate(F.interpolate(z, scale_factor=2, mode="bilinear", align_corners=True), scale_factor=2, mode="bilinear",
                                 align_corners=True)
        z = conv.conv2d(x, self.out_conv.weight, self.out_conv.bias)
        return fused.upsample2x(fused.upsample2x(z))


class ResUAPS(nn.Module):
    """Shared ResNet encoder, one clean and `n_aux` perturbed decoders; returns (main, aux1, ...) logits [B, class_num, H, W]
    like UNet_UAPS (UAPS_unet.py:224-233), so the loss block, trainer and evaluation path apply unchanged."""

    def __init__(self, in_chns: int, class_num: int, n_aux: int = 3, backbone: str = "resnet50"):
        super().__init__()
        makers = {"resnet50": resnet50, "resnet101": resnet101, "resnet152": resnet152}
        if backbone not in makers:
            raise ValueError(f"backbone {backbone!r}: the dilated Bottleneck nets {sorted(makers)} are supported")
        if not 0 <= n_aux <= 7:
            raise ValueError("n_aux must be in 0..7")
        self.n_aux = n_aux
        self.encoder = makers[backbone](in_chns)
        self.main_decoder = ResDecoder(self.encoder.channels, class_num)
        for i in range(1, n_aux + 1):
            setattr(self, f"aux_decoder{i}", ResDecoder(self.encoder.channels, class_num))
        self._noise = perturb.FeatureNoise()
        self._conv_weights = None

    def aux_decoders(self) -> List[ResDecoder]:
        return [getattr(self, f"aux_decoder{i}") for i in range(1, self.n_aux + 1)]

    def forward_pair(self, x_a, x_b):
        if x_a.shape != x_b.shape:
            raise ValueError("forward_pair: the two batches must have the same shape")
        with fused.stat_groups(2):
            return self.forward(fused.cat_batches(x_a, x_b), _groups=2)

    def forward(self, x, _groups: int = 1):
        if x.shape[2] % 8 or x.shape[3] % 8:
            raise ValueError("ResUAPS: height and width must be multiples of 8")
        if x.is_cuda:
            if self._conv_weights is None:
                self._conv_weights = [m.weight for m in self.modules() if isinstance(m, nn.Conv2d) and m.kernel_size[0] in (1, 3)]
            conv.pack_all(self._conv_weights)
            if self.training:                        # Samuelson bounds of the train-mode BatchNorm outputs (conv mode 'h16')
                if getattr(self, "_bns", None) is None:
                    self._bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
                bounds.refresh(self._bns)
        feats = self.encoder.base_forward(x)
        kinds = [_PERTURBATIONS[i % 3] for i in range(self.n_aux)]
        if x.is_cuda and self.n_aux > 0:
            fans = [perturb.perturbed_fan_out(f, kinds, _groups, self._noise.uniform_range) for f in feats]
            per_dec = [[fan[d] for fan in fans] for d in range(1 + self.n_aux)]
        else:
            per_dec = [list(feats)]
            for k in kinds:
                if k == "noise":
                    per_dec.append([f * (1 + torch.empty_like(f[0]).uniform_(-0.3, 0.3)) for f in feats])
                elif k == "dropout":
                    per_dec.append([F.dropout(f, 0.5, True) for f in feats])
                else:
                    per_dec.append([_cpu_feature_dropout(f) for f in feats])
        outs = [self.main_decoder(per_dec[0])] + [dec(per_dec[i + 1]) for i, dec in enumerate(self.aux_decoders())]
        return tuple(outs)


def _cpu_feature_dropout(x):
    """UAPS_unet.py:161-169 in plain torch (CPU inspection path of ResUAPS only)."""
    import numpy as np
    att = torch.mean(x, dim=1, keepdim=True)
    mx, _ = torch.max(att.view(x.size(0), -1), dim=1, keepdim=True)
    thr = (mx * np.random.uniform(0.7, 0.9)).view(x.size(0), 1, 1, 1).expand_as(att)
    return x.mul((att < thr).float())
