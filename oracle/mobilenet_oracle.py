"""CPU oracle for the reference's factory name `MobileNet` (TEST INFRASTRUCTURE ONLY): `DeepLabV3MobileNet`,
/root/reference/source/scripts/nets.py:277-311 = torchvision's `deeplabv3_mobilenet_v3_large` with `classifier[4]` replaced by
Conv2d(256, num_classes, 1) (:290-291); forward casts to float and returns d['out'] (:300-304); factory at
/root/reference/source/scripts/utils.py:183-184.

PARITY UNPINNED: torchvision is absent from this image and the reference's wrapper fetches COCO weights, so no
reference output can pin this model.  This file restates the PUBLISHED architecture in plain torch:
  * backbone = `mobilenet_v3_large(dilated=True).features` (Howard et al. 2019, table 1; torchvision.models.mobilenetv3): 3x3 / stride 2
    stem (16, hardswish), fifteen inverted-residual blocks (expand 1x1 -> depthwise k x k -> squeeze-excite -> linear 1x1 projection,
    residual when stride 1 and equal widths), 1x1 conv to 960 (hardswish).  `dilated=True`: the last three blocks (the "C4" stage) keep
    stride 1 and run their 5x5 depthwise convolutions at dilation 2, i.e. output stride 16.  Depthwise padding = (k - 1) // 2 * dilation.
    Every BatchNorm2d of the backbone has eps 0.001, momentum 0.01.  Squeeze-excite: global average -> 1x1 conv (bias) to
    make_divisible(expanded / 4, 8) -> ReLU -> 1x1 conv (bias) -> hardsigmoid -> per-(image, channel) scale;
  * head = torchvision's DeepLabHead(960, NC): ASPP at rates 12/24/36 (1x1 branch, three atrous 3x3, image pooling, 1x1 projection,
    Dropout(0.5) in train mode - the keep mask is an INPUT here), 3x3 conv 256 -> BN -> ReLU, 1x1 classifier, bilinear resize
    (align_corners=False) to the input size.  These BatchNorms use torch's defaults (eps 1e-5, momentum 0.1).
Parameter names: torchvision's, with `encoder.` for `backbone.` and the head names of oracle/deeplab_oracle.py (`aspp.*`, `head.*`,
`classifier`); cvcs_amd.nets.DeepLabV3MobileNet maps them to the reference wrapper's `model.backbone.*` / `model.classifier.*`.
The input is fed raw (0..255): the reference's wrapper applies no normalisation (S/nets.py:300-304).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import resnet_unet_oracle as R
from . import unet_oracle as U

BN_EPS, BN_MOMENTUM = 1e-3, 0.01
RATES = (12, 24, 36)
OUTPUT_STRIDE = 16


def make_divisible(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


# (input, kernel, expanded, out, squeeze-excite, activation, stride, dilation) - torchvision's `_mobilenet_v3_conf("mobilenet_v3_large", dilated=True)`
BLOCKS = (
    (16, 3, 16, 16, False, "RE", 1, 1),
    (16, 3, 64, 24, False, "RE", 2, 1),
    (24, 3, 72, 24, False, "RE", 1, 1),
    (24, 5, 72, 40, True, "RE", 2, 1),
    (40, 5, 120, 40, True, "RE", 1, 1),
    (40, 5, 120, 40, True, "RE", 1, 1),
    (40, 3, 240, 80, False, "HS", 2, 1),
    (80, 3, 200, 80, False, "HS", 1, 1),
    (80, 3, 184, 80, False, "HS", 1, 1),
    (80, 3, 184, 80, False, "HS", 1, 1),
    (80, 3, 480, 112, True, "HS", 1, 1),
    (112, 3, 672, 112, True, "HS", 1, 1),
    (112, 5, 672, 160, True, "HS", 2, 2),
    (160, 5, 960, 160, True, "HS", 1, 2),
    (160, 5, 960, 160, True, "HS", 1, 2),
)
LAST_CHANNELS = 960


def block_layers(i):
    """the sub-module indices torchvision's InvertedResidual gives the layers of block i (0-based in BLOCKS; module `features[i + 1]`):
    (expand | None, depthwise, squeeze-excite | None, project)"""
    cin, k, exp, cout, se, act, stride, dil = BLOCKS[i]
    j = 0
    e = None
    if exp != cin:
        e, j = j, j + 1
    d, j = j, j + 1
    s = None
    if se:
        s, j = j, j + 1
    return e, d, s, j


def param_spec(num_classes):
    spec = []

    def conv(p, cin, cout, k, bias=False, groups=1):
        spec.append((p + ".weight", (cout, cin // groups, k, k)))
        if bias:
            spec.append((p + ".bias", (cout,)))

    def bn(p, c):
        spec.extend([(p + ".weight", (c,)), (p + ".bias", (c,)), (p + ".running_mean", (c,)), (p + ".running_var", (c,))])

    conv("encoder.0.0", 3, 16, 3); bn("encoder.0.1", 16)
    for i, (cin, k, exp, cout, se, act, stride, dil) in enumerate(BLOCKS):
        p = f"encoder.{i + 1}.block"
        e, d, s, pr = block_layers(i)
        if e is not None:
            conv(f"{p}.{e}.0", cin, exp, 1); bn(f"{p}.{e}.1", exp)
        conv(f"{p}.{d}.0", exp, exp, k, groups=exp); bn(f"{p}.{d}.1", exp)
        if s is not None:
            sq = make_divisible(exp // 4, 8)
            conv(f"{p}.{s}.fc1", exp, sq, 1, bias=True)
            conv(f"{p}.{s}.fc2", sq, exp, 1, bias=True)
        conv(f"{p}.{pr}.0", exp, cout, 1); bn(f"{p}.{pr}.1", cout)
    conv(f"encoder.{len(BLOCKS) + 1}.0", BLOCKS[-1][3], LAST_CHANNELS, 1); bn(f"encoder.{len(BLOCKS) + 1}.1", LAST_CHANNELS)
    cin = LAST_CHANNELS
    conv("aspp.convs.0.0", cin, 256, 1); bn("aspp.convs.0.1", 256)
    for i in (1, 2, 3):
        conv(f"aspp.convs.{i}.0", cin, 256, 3); bn(f"aspp.convs.{i}.1", 256)
    conv("aspp.convs.4.1", cin, 256, 1); bn("aspp.convs.4.2", 256)
    conv("aspp.project.0", 5 * 256, 256, 1); bn("aspp.project.1", 256)
    conv("head.0", 256, 256, 3); bn("head.1", 256)
    conv("classifier", 256, num_classes, 1, bias=True)
    return spec


is_buffer = R.is_buffer


def init_params(num_classes, seed=0):
    """random weights at a scale that keeps every layer's activations O(1) on raw 0..255 inputs (the first BatchNorm absorbs the input range)"""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for name, shape in param_spec(num_classes):
        if name.endswith("running_mean"):
            out[name] = torch.zeros(shape)
        elif name.endswith("running_var"):
            out[name] = torch.ones(shape)
        elif len(shape) == 4:
            out[name] = (torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[1] * shape[2] * shape[3]))).float()
        elif name.endswith(".weight"):            # BatchNorm gamma
            out[name] = (0.5 + torch.rand(shape, generator=g)).float()
        else:                                     # BatchNorm beta, conv biases
            out[name] = ((torch.rand(shape, generator=g) - 0.5) * 0.2).float()
    return out


def _bn(x, p, prefix, train, backbone):
    eps, mom = (BN_EPS, BN_MOMENTUM) if backbone else (R.BN_EPS, R.BN_MOMENTUM)
    return F.batch_norm(x, p[prefix + ".running_mean"], p[prefix + ".running_var"], p[prefix + ".weight"], p[prefix + ".bias"], training=train,
                        momentum=mom, eps=eps)


def _act(y, act):
    return F.relu(y) if act == "RE" else F.hardswish(y) if act == "HS" else y


def _cna(x, p, prefix, act, train, q, stride=1, pad=0, dil=1, groups=1):
    """torchvision's Conv2dNormActivation: conv (no bias) -> BatchNorm(eps 1e-3, momentum 0.01) -> activation"""
    y = R._q(F.conv2d(x, R._qw(p[prefix + ".0.weight"], q), None, stride=stride, padding=pad, dilation=dil, groups=groups), q)
    return R._q(_act(_bn(y, p, prefix + ".1", train, True), act), q)


def backbone(p, x, train=False, emulate_bf16=False, return_blocks=False):
    q = emulate_bf16
    h = _cna(x, p, "encoder.0", "HS", train, q, stride=2, pad=1)
    outs = [h]
    for i, (cin, k, exp, cout, se, act, stride, dil) in enumerate(BLOCKS):
        pre = f"encoder.{i + 1}.block"
        e, d, s, pr = block_layers(i)
        t = h
        if e is not None:
            t = _cna(t, p, f"{pre}.{e}", act, train, q)
        st = 1 if dil > 1 else stride            # (torchvision: a dilated block keeps stride 1)
        t = _cna(t, p, f"{pre}.{d}", act, train, q, stride=st, pad=(k - 1) // 2 * dil, dil=dil, groups=exp)
        if s is not None:
            z = R._q(t.mean(dim=(2, 3), keepdim=True), q)
            z = R._q(F.relu(F.conv2d(z, R._qw(p[f"{pre}.{s}.fc1.weight"], q), p[f"{pre}.{s}.fc1.bias"])), q)
            z = R._q(F.hardsigmoid(R._q(F.conv2d(z, R._qw(p[f"{pre}.{s}.fc2.weight"], q), p[f"{pre}.{s}.fc2.bias"]), q)), q)
            t = R._q(t * z, q)
        t = _cna(t, p, f"{pre}.{pr}", None, train, q)
        if stride == 1 and cin == cout:
            t = R._q(t + h, q)
        h = t
        outs.append(h)
    h = _cna(h, p, f"encoder.{len(BLOCKS) + 1}", "HS", train, q)
    outs.append(h)
    return (h, outs) if return_blocks else h


def _cbr(x, p, conv, bn, pad, q, train, dil=1):
    y = R._q(F.conv2d(x, R._qw(p[conv + ".weight"], q), None, padding=pad, dilation=dil), q)
    return R._q(F.relu(_bn(y, p, bn, train, False)), q)


def forward(p, x, train=False, emulate_bf16=False, return_acts=False, aspp_keep=None, aspp_dropout=0.5):
    """x f32 [B,3,S,S] raw 0..255 (S % 32 == 0) -> logits f32 [B,NC,S,S]"""
    q = emulate_bf16
    f5, blocks = backbone(p, R._q(x, q), train, q, return_blocks=True)
    branches = [_cbr(f5, p, "aspp.convs.0.0", "aspp.convs.0.1", 0, q, train)]
    for i, r in enumerate(RATES, start=1):
        branches.append(_cbr(f5, p, f"aspp.convs.{i}.0", f"aspp.convs.{i}.1", r, q, train, dil=r))
    g = R._q(f5.mean(dim=(2, 3), keepdim=True), q)
    g = _cbr(g, p, "aspp.convs.4.1", "aspp.convs.4.2", 0, q, train)
    branches.append(g.expand(-1, -1, f5.shape[2], f5.shape[3]))
    a = _cbr(torch.cat(branches, dim=1), p, "aspp.project.0", "aspp.project.1", 0, q, train)
    if aspp_keep is not None:
        a = R._q(a * aspp_keep.to(a.dtype) * (1.0 / (1.0 - aspp_dropout)), q)
    d = _cbr(a, p, "head.0", "head.1", 1, q, train)
    z = F.conv2d(d, p["classifier.weight"], p["classifier.bias"])
    logits = F.interpolate(z, scale_factor=OUTPUT_STRIDE, mode="bilinear", align_corners=False)
    return (logits, {"blocks": blocks, "features": f5, "aspp": a, "last": d}) if return_acts else logits


class OracleTrainer(R.OracleTrainer):
    """forward -> loss -> zero_grad -> backward -> opt.step (S/train.py:121-126) on the definition above"""

    def __init__(self, num_classes, opt="SGD2", epochs=20, ignore_index=-100, weight=None, seed=0, params=None, emulate_bf16=False):
        super().__init__("mobilenet_v3_large", num_classes, opt, epochs, ignore_index, weight, seed,
                         params if params is not None else init_params(num_classes, seed), emulate_bf16)

    def step(self, x_u8, target_u8):
        logits = forward(self.p, x_u8.to(torch.float32), train=True, emulate_bf16=self.emulate_bf16)
        loss = U.cross_entropy(logits, target_u8.long(), self.weight, self.ignore_index)
        grads = torch.autograd.grad(loss, [self.p[k] for k in self.trainable])
        self.nstep += 1
        lr = self.lr()
        with torch.no_grad():
            for k, g in zip(self.trainable, grads):
                p = self.p[k]
                if self.cfg["kind"] == "sgd":
                    newp, buf = U.sgd_step(p, g, self.state.get(k), lr, self.cfg["momentum"], self.cfg["weight_decay"], k not in self.state)
                    self.state[k] = buf
                else:
                    m, v = self.state.get(k, (torch.zeros_like(p), torch.zeros_like(p)))
                    newp, m, v = U.adam_step(p, g, m, v, self.nstep, lr, self.cfg["betas"], self.cfg["eps"], self.cfg["weight_decay"])
                    self.state[k] = (m, v)
                p.copy_(newp)
        return float(loss.detach()), logits.detach(), dict(zip(self.trainable, grads))
