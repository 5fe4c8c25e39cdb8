"""CPU oracle for the DeepLab family (TEST INFRASTRUCTURE ONLY): DeepLabV3+ of BASELINE.json configs[2] ("DeepLabV3+
(ResNet50, ASPP dilations 6/12/18)") and the DeepLabV3 heads the reference wraps from torchvision (`Resnet101`,
S/nets.py:234-257: `deeplabv3_resnet101` with `classifier[4]` replaced by Conv2d(256, num_classes, 1), forward returns
d['out'] :248-250).

PARITY UNPINNED: torchvision is absent from this image and the reference's wrappers fetch pretrained
weights, so no reference output can pin these models.  This file restates the PUBLISHED architectures in plain torch:
  * encoder: ResNet v1.5 (oracle/resnet_unet_oracle.py, cross-checked against transformers.ResNetModel) with torchvision's
    `replace_stride_with_dilation` rule (output stride 16: last stage dilated; 8: last two);
  * ASPP (Chen et al. 2017, torchvision.models.segmentation.deeplabv3.ASPP): 1x1 branch, three 3x3 atrous branches (rates
    6/12/18 at OS 16; 12/24/36 at OS 8), image pooling branch (AdaptiveAvgPool2d(1) -> 1x1 conv -> BN -> ReLU -> bilinear
    resize = broadcast), concat -> 1x1 projection -> BN -> ReLU -> Dropout(0.5) in train mode: `forward(..., aspp_keep=mask)`
    takes the keep mask as an input (the HIP path's mask is a counter-based hash the tests replay on the host);
  * DeepLabV3 head (plus=False; torchvision DeepLabHead): ASPP -> 3x3 conv 256 -> BN -> ReLU -> 1x1 classifier -> bilinear
    resize (align_corners=False) to the input size;
  * DeepLabV3+ decoder (plus=True; Chen et al. 2018): 1x1 reduction of the stride-4 feature (64 channels here instead of the
    paper's 48: the MFMA kernels tile output channels by 64) -> BN -> ReLU; ASPP output x4 bilinear; concat; two
    (3x3 conv 256 -> BN -> ReLU); 1x1 classifier at stride 4; x4 bilinear resize to the input size.
Inputs are the raw 0..255 tile values (S/train.py:121).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import this file.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import resnet_unet_oracle as R
from . import unet_oracle as U

LOW_LEVEL_CHANNELS = 64


def rates_for(output_stride):
    return (6, 12, 18) if output_stride == 16 else (12, 24, 36)


def param_spec(arch, num_classes, plus=True):
    kind, depths, widths = R.ARCHS[arch]
    spec = [(k, s) for k, s in R.param_spec(arch, num_classes) if k.startswith("encoder.")]

    def conv(p, cin, cout, k, bias=False):
        spec.append((p + ".weight", (cout, cin, k, k)))
        if bias:
            spec.append((p + ".bias", (cout,)))

    def bn(p, c):
        spec.extend([(p + ".weight", (c,)), (p + ".bias", (c,)), (p + ".running_mean", (c,)), (p + ".running_var", (c,))])

    cin = widths[3]
    conv("aspp.convs.0.0", cin, 256, 1); bn("aspp.convs.0.1", 256)
    for i in (1, 2, 3):
        conv(f"aspp.convs.{i}.0", cin, 256, 3); bn(f"aspp.convs.{i}.1", 256)
    conv("aspp.convs.4.1", cin, 256, 1); bn("aspp.convs.4.2", 256)
    conv("aspp.project.0", 5 * 256, 256, 1); bn("aspp.project.1", 256)
    if plus:
        conv("low_level.0", widths[0], LOW_LEVEL_CHANNELS, 1); bn("low_level.1", LOW_LEVEL_CHANNELS)
        conv("decoder.conv1.0", 256 + LOW_LEVEL_CHANNELS, 256, 3); bn("decoder.conv1.1", 256)
        conv("decoder.conv2.0", 256, 256, 3); bn("decoder.conv2.1", 256)
    else:
        conv("head.0", 256, 256, 3); bn("head.1", 256)
    conv("classifier", 256, num_classes, 1, bias=True)
    return spec


def init_params(arch, num_classes, seed=0, plus=True):
    g = torch.Generator().manual_seed(seed)
    enc = R.init_params(arch, num_classes, seed)
    out = OrderedDict()
    for name, shape in param_spec(arch, num_classes, plus):
        if name in enc:
            out[name] = enc[name]
        elif name.endswith("running_mean"):
            out[name] = torch.zeros(shape)
        elif name.endswith("running_var"):
            out[name] = torch.ones(shape)
        elif len(shape) == 4:
            out[name] = (torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[1] * shape[2] * shape[3]))).float()
        elif name == "classifier.bias":
            out[name] = ((torch.rand(shape, generator=g) - 0.5) * 0.2).float()
        elif name.endswith(".weight"):
            out[name] = (0.5 + torch.rand(shape, generator=g)).float()
        else:
            out[name] = ((torch.rand(shape, generator=g) - 0.5) * 0.2).float()
    return out


is_buffer = R.is_buffer


def forward(p, x, arch="resnet50", train=False, output_stride=16, plus=True, emulate_bf16=False, return_acts=False, aspp_keep=None,
            aspp_dropout=0.5):
    """x f32 [B,3,S,S] raw 0..255 (S % 32 == 0) -> logits f32 [B,NC,S,S].  aspp_keep: bool [B,256,h,w] - the keep mask of the
    nn.Dropout(aspp_dropout) torchvision's ASPP ends with (train mode); None: no dropout (eval mode, or the deterministic parity runs).
    The mask itself is an INPUT here: the HIP path draws it from a counter-based hash that the tests replay on the host."""
    q = emulate_bf16
    feats = R.encoder_features(p, x, arch, train, q, output_stride)
    f5 = feats[4]
    branches = [R._cbr(f5, p, "aspp.convs.0.0", "aspp.convs.0.1", 1, 0, q, train)]
    for i, r in enumerate(rates_for(output_stride), start=1):
        branches.append(R._cbr(f5, p, f"aspp.convs.{i}.0", f"aspp.convs.{i}.1", 1, r, q, train, dil=r))
    g = R._q(f5.mean(dim=(2, 3), keepdim=True), q)
    g = R._cbr(g, p, "aspp.convs.4.1", "aspp.convs.4.2", 1, 0, q, train)
    branches.append(g.expand(-1, -1, f5.shape[2], f5.shape[3]))        # bilinear resize of a 1x1 map = broadcast
    a = R._cbr(torch.cat(branches, dim=1), p, "aspp.project.0", "aspp.project.1", 1, 0, q, train)
    if aspp_keep is not None:
        a = R._q(a * aspp_keep.to(a.dtype) * (1.0 / (1.0 - aspp_dropout)), q)
    acts = {"features": feats, "aspp": a}
    if plus:
        low = R._cbr(feats[1], p, "low_level.0", "low_level.1", 1, 0, q, train)
        up = R._q(F.interpolate(a, scale_factor=output_stride // 4, mode="bilinear", align_corners=False), q)
        d = R._cbr(torch.cat([up, low], dim=1), p, "decoder.conv1.0", "decoder.conv1.1", 1, 1, q, train)
        d = R._cbr(d, p, "decoder.conv2.0", "decoder.conv2.1", 1, 1, q, train)
        final = 4
    else:
        d = R._cbr(a, p, "head.0", "head.1", 1, 1, q, train)
        final = output_stride
    acts["last"] = d
    z = F.conv2d(d, p["classifier.weight"], p["classifier.bias"])
    logits = F.interpolate(z, scale_factor=final, mode="bilinear", align_corners=False)
    return (logits, acts) if return_acts else logits


class OracleTrainer(R.OracleTrainer):
    """forward -> loss -> zero_grad -> backward -> opt.step (S/train.py:121-126) on the DeepLab definition above"""

    def __init__(self, arch, num_classes, opt="SGD2", epochs=20, ignore_index=-100, weight=None, seed=0, params=None,
                 emulate_bf16=False, output_stride=16, plus=True):
        self.os, self.plus = output_stride, plus
        super().__init__(arch, num_classes, opt, epochs, ignore_index, weight, seed,
                         params if params is not None else init_params(arch, num_classes, seed, plus), emulate_bf16)

    def step(self, x_u8, target_u8):
        logits = forward(self.p, x_u8.to(torch.float32), self.arch, train=True, output_stride=self.os, plus=self.plus,
                         emulate_bf16=self.emulate_bf16)
        loss = U.cross_entropy(logits, target_u8.long(), self.weight, self.ignore_index)
        grads = torch.autograd.grad(loss, [self.p[k] for k in self.trainable])
        self.nstep += 1
        lr = self.lr()
        with torch.no_grad():
            for k, g in zip(self.trainable, grads):
                p = self.p[k]
                if self.cfg["kind"] == "sgd":
                    newp, buf = U.sgd_step(p, g, self.state.get(k), lr, self.cfg["momentum"], self.cfg["weight_decay"],
                                           k not in self.state)
                    self.state[k] = buf
                else:
                    m, v = self.state.get(k, (torch.zeros_like(p), torch.zeros_like(p)))
                    newp, m, v = U.adam_step(p, g, m, v, self.nstep, lr, self.cfg["betas"], self.cfg["eps"], self.cfg["weight_decay"])
                    self.state[k] = (m, v)
                p.copy_(newp)
        return float(loss.detach()), logits.detach(), dict(zip(self.trainable, grads))
