"""CPU oracle for the ResNet-encoder U-Nets of BASELINE.json's configs 1, 2 and 5 (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: the reference (theElandor/CVCS) has NO ResNet-encoder U-Net - `S/utils.py:174-195` (`load_network`)
knows `Unet`, `Unetv2`, `Resnet101`, `MobileNet`, `Ensemble`, `SegformerMod` only - so no reference
output can pin this model.  BASELINE.json names "ResNet18-UNet" / "ResNet50-UNet" as targets of the new build; this file
is the build's own plain torch-CPU fp32 definition of them and the checker of the HIP path:

  * encoder = the standard ResNet v1.5 (He et al. 2015; stride on the 3x3 of a bottleneck): 7x7/s2 stem conv (no bias)
    -> BN -> ReLU -> MaxPool 3x3/s2/p1 -> four stages of BasicBlock (resnet18: 2,2,2,2) or Bottleneck (resnet50:
    3,4,6,3) blocks.  `tests/test_resnet_oracle_cpu.py` cross-checks it against `transformers.ResNetModel` built from a
    LOCAL config (no fetch) with the same weights: that is an independent implementation of the
    same published architecture, not the reference.
  * decoder = U-Net decoder in the north star's words ("bilinear-upsample decoder"): five stages of
    bilinear x2 (`nn.Upsample(scale_factor=2, mode='bilinear')`, align_corners=False: the reference's own up-sampling op,
    S/blocks.py:29) -> concat with the encoder feature of that resolution (none at full resolution) ->
    (conv3x3 no-bias -> BN -> ReLU) x 2 (the reference's UnetEncodeLayer order conv -> BN -> ReLU, S/blocks.py:8-23),
    widths (256, 128, 64, 64, 64); head = 1x1 conv to num_classes with bias (S/nets.py:172 style).
  * inputs are the raw 0..255 tile values cast to f32, un-normalised, exactly as the reference feeds its own networks
    (S/train.py:121).
  * loss / optimiser / step order are the reference's (`oracle/unet_oracle.py`: CE S/utils.py:230, SGD2/ADAM1
    S/utils.py:208-221, step order S/train.py:121-126).

`emulate_bf16=True` rounds to bfloat16 exactly where the HIP bf16 path stores or feeds bf16 (conv operands, every stored
activation and every stored activation gradient; accumulation, BatchNorm statistics and parameters stay f32): the bf16
HIP path is held against THIS mode with a tight tolerance, which separates rounding from bugs.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import unet_oracle as U

BN_EPS, BN_MOMENTUM = U.BN_EPS, U.BN_MOMENTUM

ARCHS = {
    # name: (block kind, blocks per stage, stage output widths, bottleneck mid widths)
    "resnet18": ("basic", (2, 2, 2, 2), (64, 128, 256, 512)),
    "resnet34": ("basic", (3, 4, 6, 3), (64, 128, 256, 512)),
    "resnet50": ("bottleneck", (3, 4, 6, 3), (256, 512, 1024, 2048)),
    "resnet101": ("bottleneck", (3, 4, 23, 3), (256, 512, 1024, 2048)),
}
DECODER_CHANNELS = (256, 128, 64, 32, 16)         # the common public layout (segmentation_models_pytorch's Unet default)
GN_DECODER_CHANNELS = (256, 128, 64, 64, 64)      # decoder_norm="gn_silu": GroupNorm(32) needs multiples of 32 channels


def _dec_channels(decoder_channels, decoder_norm="bn_relu"):
    if decoder_channels is not None:
        return tuple(decoder_channels)
    return DECODER_CHANNELS if decoder_norm == "bn_relu" else GN_DECODER_CHANNELS


# --------------------------------------------------------------------------- parameter table
GN_GROUPS = 32


def param_spec(arch: str, num_classes: int, decoder_channels=None, decoder_norm="bn_relu"):
    """ordered (name, shape); encoder names are torchvision's ResNet names under `encoder.`, so ImageNet / torchvision
    checkpoints map one to one.  decoder_norm="gn_silu": the decoder's conv -> BatchNorm -> ReLU pairs become
    conv -> GroupNorm(32) -> SiLU (north star "fused BN/GN+SiLU"); a GroupNorm has no running statistics."""
    kind, depths, widths = ARCHS[arch]
    spec = []

    def conv(p, cin, cout, k, bias=False):
        spec.append((p + ".weight", (cout, cin, k, k)))
        if bias:
            spec.append((p + ".bias", (cout,)))

    def bn(p, c):
        spec.extend([(p + ".weight", (c,)), (p + ".bias", (c,)), (p + ".running_mean", (c,)), (p + ".running_var", (c,))])

    def dnorm(p, c):
        if decoder_norm == "gn_silu":
            spec.extend([(p + ".weight", (c,)), (p + ".bias", (c,))])
        else:
            bn(p, c)

    conv("encoder.conv1", 3, 64, 7)
    bn("encoder.bn1", 64)
    cin = 64
    for s, (n, w) in enumerate(zip(depths, widths), start=1):
        for b in range(n):
            p = f"encoder.layer{s}.{b}"
            stride = 2 if (b == 0 and s > 1) else 1
            if kind == "basic":
                conv(p + ".conv1", cin, w, 3); bn(p + ".bn1", w)
                conv(p + ".conv2", w, w, 3); bn(p + ".bn2", w)
            else:
                mid = w // 4
                conv(p + ".conv1", cin, mid, 1); bn(p + ".bn1", mid)
                conv(p + ".conv2", mid, mid, 3); bn(p + ".bn2", mid)
                conv(p + ".conv3", mid, w, 1); bn(p + ".bn3", w)
            if stride != 1 or cin != w:
                conv(p + ".downsample.0", cin, w, 1); bn(p + ".downsample.1", w)
            cin = w
    skips = [widths[2], widths[1], widths[0], 64, 0]
    for i, (dc, sk) in enumerate(zip(_dec_channels(decoder_channels, decoder_norm), skips)):
        p = f"decoder.blocks.{i}"
        conv(p + ".conv1.0", cin + sk, dc, 3); dnorm(p + ".conv1.1", dc)
        conv(p + ".conv2.0", dc, dc, 3); dnorm(p + ".conv2.1", dc)
        cin = dc
    conv("segmentation_head.0", cin, num_classes, 1, bias=True)
    return spec


def is_buffer(name):
    return name.endswith("running_mean") or name.endswith("running_var")


def init_params(arch, num_classes, seed=0, decoder_channels=None, decoder_norm="bn_relu"):
    """deterministic synthetic weights, independent of torch's default init: conv ~ N(0, 2/fan_in); BN gamma in [0.5,1.5]
    (the last BN of every residual block in [0.2,0.6] so that 16 stacked blocks keep O(1) activations), beta small."""
    g = torch.Generator().manual_seed(seed)
    kind = ARCHS[arch][0]
    last_bn = ".bn3.weight" if kind == "bottleneck" else ".bn2.weight"
    out = OrderedDict()
    for name, shape in param_spec(arch, num_classes, decoder_channels, decoder_norm):
        if name.endswith("running_mean"):
            t = torch.zeros(shape)
        elif name.endswith("running_var"):
            t = torch.ones(shape)
        elif len(shape) == 4:
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[1] * shape[2] * shape[3]))
        elif name == "segmentation_head.0.bias":
            t = (torch.rand(shape, generator=g) - 0.5) * 0.2
        elif name.endswith(".weight"):
            t = (0.2 + 0.4 * torch.rand(shape, generator=g)) if name.endswith(last_bn) else (0.5 + torch.rand(shape, generator=g))
        else:
            t = (torch.rand(shape, generator=g) - 0.5) * 0.2
        out[name] = t.float()
    return out


# --------------------------------------------------------------------------- bf16 emulation
class _RoundBF16(torch.autograd.Function):
    """value rounded to bfloat16 (round-to-nearest-even) forward; the gradient passing back is rounded likewise - the HIP
    path stores both the activation and its gradient in bf16."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class _RoundFwdOnly(torch.autograd.Function):
    """weights: the packed MFMA operand is bf16, the weight gradient is accumulated and kept in f32"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def _q(x, on):
    return _RoundBF16.apply(x) if on else x


def _qw(w, on):
    return _RoundFwdOnly.apply(w) if on else w


# --------------------------------------------------------------------------- forward
def _bn(x, p, prefix, train):
    rm, rv = p[prefix + ".running_mean"], p[prefix + ".running_var"]
    return F.batch_norm(x, rm, rv, p[prefix + ".weight"], p[prefix + ".bias"], training=train, momentum=BN_MOMENTUM,
                        eps=BN_EPS)


# --------------------------------------------------------------------------- fp8 emulation (BASELINE configs[4]: "mixed bf16/fp8 convs")
_FP8_BOOK = None      # oracle.fp8_emulation.ScaleBook while forward(..., emulate_fp8=book) runs, else None


def fp8_eligible(w, x, stride, pad, dil):
    """the layers cvcs_amd.resnet_engine._fp8_ok sends to the fp8 kernel: 3x3 / stride 1 / pad 1, both channel counts multiples of 128,
    maps of at least 8 pixels"""
    return (w.shape[2] == 3 and w.shape[3] == 3 and stride == 1 and pad == 1 and dil == 1 and w.shape[0] % 128 == 0 and w.shape[1] % 128 == 0 and
            x.shape[2] >= 8 and x.shape[3] >= 8)


class _Fp8Conv(torch.autograd.Function):
    """3x3 convolution as the MI355X fp8 path computes it: forward e4m3(x / s_x) * e4m3(bf16(w) / s_w), products exact, f32 accumulate;
    data gradient e5m2(dy / s_g) against the SAME e4m3 weights; the weight gradient stays on the bf16 operands (x, dy as stored)."""

    @staticmethod
    def forward(ctx, x, w, name):
        from oracle import fp8_emulation as F8
        book = _FP8_BOOK
        wb = w.to(torch.bfloat16).to(torch.float32)
        sx = book.scale("a:" + name, x, F8.E4M3)
        sw = book.scale("w:" + name, wb, F8.E4M3)
        xq, wq = F8.fake_quant(x, sx, F8.E4M3), F8.fake_quant(wb, sw, F8.E4M3)
        ctx.save_for_backward(x, wq)
        ctx.name, ctx.book = name, book
        return F.conv2d(xq, wq, padding=1)

    @staticmethod
    def backward(ctx, g):
        from oracle import fp8_emulation as F8
        x, wq = ctx.saved_tensors
        sg = ctx.book.scale("g:" + ctx.name, g, F8.E5M2)
        gq = F8.fake_quant(g, sg, F8.E5M2)
        dx = torch.nn.grad.conv2d_input(x.shape, wq, gq, padding=1)
        dw = torch.nn.grad.conv2d_weight(x, wq.shape, g, padding=1)
        return dx, dw, None


def _conv(x, p, name, stride=1, pad=0, q=False, dil=1):
    # x is already bf16-representable in emulation mode (every stored activation is rounded where it is produced)
    w = p[name + ".weight"]
    if _FP8_BOOK is not None and fp8_eligible(w, x, stride, pad, dil):
        return _q(_Fp8Conv.apply(x, w, name), True)
    return _q(F.conv2d(x, _qw(w, q), p.get(name + ".bias"), stride=stride, padding=pad, dilation=dil), q)


def _cbr(x, p, conv, bn, stride, pad, q, train, relu=True, dil=1):
    y = _bn(_conv(x, p, conv, stride, pad, q, dil), p, bn, train)
    return _q(F.relu(y), q) if relu else y


def _block(x, p, prefix, kind, stride, q, train, dil=1):
    """dil: dilation (= padding) of the block's 3x3 convolution(s) - the dilated stages of the DeepLab encoders"""
    if kind == "basic":
        y = _cbr(x, p, prefix + ".conv1", prefix + ".bn1", stride, dil, q, train, dil=dil)
        z = _cbr(y, p, prefix + ".conv2", prefix + ".bn2", 1, dil, q, train, relu=False, dil=dil)
    else:
        y = _cbr(x, p, prefix + ".conv1", prefix + ".bn1", 1, 0, q, train)
        y = _cbr(y, p, prefix + ".conv2", prefix + ".bn2", stride, dil, q, train, dil=dil)
        z = _cbr(y, p, prefix + ".conv3", prefix + ".bn3", 1, 0, q, train, relu=False)
    if (prefix + ".downsample.0.weight") in p:
        idt = _cbr(x, p, prefix + ".downsample.0", prefix + ".downsample.1", stride, 0, q, train, relu=False)
    else:
        idt = x
    return _q(F.relu(z + idt), q)


def stage_plan(arch, output_stride=32):
    """[(stage, block, stride, dilation)]: torchvision's `replace_stride_with_dilation` rule - once the output stride is
    reached, a stage's stride-2 block keeps stride 1 (and the dilation so far), the stage's later blocks double the dilation"""
    _, depths, _ = ARCHS[arch]
    plan, cur, dil = [], 4, 1
    for s, n in enumerate(depths, start=1):
        first_stride, first_dil = 1, dil
        if s > 1:
            if cur >= output_stride:
                dil *= 2
            else:
                first_stride, cur = 2, cur * 2
        for b in range(n):
            plan.append((s, b, first_stride if b == 0 else 1, first_dil if b == 0 else dil))
    return plan


def encoder_features(p, x, arch, train=False, emulate_bf16=False, output_stride=32):
    """[f1 (64, S/2), f2 (S/4), f3 (S/8), f4 (S/16), f5 (S/32)]  (output_stride 16 / 8: the last one / two stages dilated)"""
    kind, depths, _ = ARCHS[arch]
    q = emulate_bf16
    x = _q(x, q)
    f1 = _cbr(x, p, "encoder.conv1", "encoder.bn1", 2, 3, q, train)
    h = F.max_pool2d(f1, 3, 2, 1)
    feats = [f1]
    for s, b, stride, dil in stage_plan(arch, output_stride):
        h = _block(h, p, f"encoder.layer{s}.{b}", kind, stride, q, train, dil)
        if b == depths[s - 1] - 1:
            feats.append(h)
    return feats


def _cgs(x, p, conv, gn, q):
    """conv3x3 -> GroupNorm(32 groups, eps 1e-5) -> SiLU (the decoder_norm="gn_silu" option)"""
    y = _conv(x, p, conv, 1, 1, q)
    return _q(F.silu(F.group_norm(y, GN_GROUPS, p[gn + ".weight"], p[gn + ".bias"], eps=BN_EPS)), q)


def forward(p, x, arch="resnet50", train=False, emulate_bf16=False, return_acts=False, emulate_fp8=None):
    """x f32 [B,3,S,S] raw 0..255 (S % 32 == 0) -> logits f32 [B,NC,S,S].  emulate_fp8: an oracle.fp8_emulation.ScaleBook - the bf16
    emulation with the qualifying 3x3 convolutions in fp8 (precision "fp8" of cvcs_amd; the caller ends the book's step)"""
    global _FP8_BOOK
    if emulate_fp8 is not None:
        prev, _FP8_BOOK = _FP8_BOOK, emulate_fp8
        try:
            return forward(p, x, arch, train, True, return_acts, None)
        finally:
            _FP8_BOOK = prev
    q = emulate_bf16
    feats = encoder_features(p, x, arch, train, q)
    skips = [feats[3], feats[2], feats[1], feats[0], None]
    h = feats[4]
    acts = {"features": feats}
    i = 0
    while f"decoder.blocks.{i}.conv1.0.weight" in p:
        h = _q(F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=False), q)
        if skips[i] is not None:
            h = torch.cat([h, skips[i]], dim=1)
        pre = f"decoder.blocks.{i}"
        if (pre + ".conv1.1.running_mean") in p:
            h = _cbr(h, p, pre + ".conv1.0", pre + ".conv1.1", 1, 1, q, train)
            h = _cbr(h, p, pre + ".conv2.0", pre + ".conv2.1", 1, 1, q, train)
        else:   # GroupNorm + SiLU decoder
            h = _cgs(h, p, pre + ".conv1.0", pre + ".conv1.1", q)
            h = _cgs(h, p, pre + ".conv2.0", pre + ".conv2.1", q)
        acts[f"decoder{i}"] = h
        i += 1
    # the head reads the stored activation and keeps f32 weights / f32 logits in both precisions
    logits = F.conv2d(h, p["segmentation_head.0.weight"], p["segmentation_head.0.bias"])
    return (logits, acts) if return_acts else logits


# --------------------------------------------------------------------------- algorithmic work (bench.py)
def conv_table(arch, num_classes, S, decoder_channels=None):
    """[(name, Cin, Cout, K, stride, Hout)] of every convolution at tile side S - the layer table behind
    `algorithmic_gflop_per_step` (FLOPs = 2*Cin*Cout*K*K*Hout*Hout per tile)."""
    kind, depths, widths = ARCHS[arch]
    rows = [("encoder.conv1", 3, 64, 7, 2, S // 2)]
    cin, h = 64, S // 4
    for s, (n, w) in enumerate(zip(depths, widths), start=1):
        for b in range(n):
            stride = 2 if (b == 0 and s > 1) else 1
            ho = h // stride
            p = f"encoder.layer{s}.{b}"
            if kind == "basic":
                rows += [(p + ".conv1", cin, w, 3, stride, ho), (p + ".conv2", w, w, 3, 1, ho)]
            else:
                mid = w // 4
                rows += [(p + ".conv1", cin, mid, 1, 1, h), (p + ".conv2", mid, mid, 3, stride, ho), (p + ".conv3", mid, w, 1, 1, ho)]
            if stride != 1 or cin != w:
                rows.append((p + ".downsample.0", cin, w, 1, stride, ho))
            cin, h = w, ho
    skips = [widths[2], widths[1], widths[0], 64, 0]
    for i, (dc, sk) in enumerate(zip(_dec_channels(decoder_channels), skips)):
        h *= 2
        rows += [(f"decoder.blocks.{i}.conv1.0", cin + sk, dc, 3, 1, h), (f"decoder.blocks.{i}.conv2.0", dc, dc, 3, 1, h)]
        cin = dc
    rows.append(("segmentation_head.0", cin, num_classes, 1, 1, h))
    return rows


def forward_gflop_per_tile(arch, num_classes, S, decoder_channels=None):
    return sum(2.0 * ci * co * k * k * ho * ho for _, ci, co, k, _, ho in conv_table(arch, num_classes, S, decoder_channels)) / 1e9


# --------------------------------------------------------------------------- one reference-order training step
class OracleTrainer:
    """forward -> loss -> zero_grad -> backward -> opt.step (S/train.py:121-126) on the ResNet-UNet definition above"""

    def __init__(self, arch, num_classes, opt="SGD2", epochs=20, ignore_index=-100, weight=None, seed=0, params=None,
                 emulate_bf16=False, decoder_norm="bn_relu", emulate_fp8=False):
        self.book = None
        if emulate_fp8:
            from oracle import fp8_emulation as F8
            self.book, emulate_bf16 = F8.ScaleBook(), True
        self.arch = arch
        self.p = params if params is not None else init_params(arch, num_classes, seed, decoder_norm=decoder_norm)
        self.trainable = [k for k in self.p if not is_buffer(k)]
        for k in self.trainable:
            self.p[k].requires_grad_(True)
        self.cfg = dict(U.OPTIMIZERS[opt])
        if self.cfg["total_iters"] is None:
            self.cfg["total_iters"] = epochs
        self.state, self.nstep, self.epoch = {}, 0, 0
        self.ignore_index, self.weight, self.emulate_bf16 = ignore_index, weight, emulate_bf16

    def lr(self):
        return U.polynomial_lr(self.cfg["lr"], self.epoch, self.cfg["total_iters"], self.cfg["power"])

    def step(self, x_u8, target_u8):
        global _FP8_BOOK
        logits = forward(self.p, x_u8.to(torch.float32), self.arch, train=True, emulate_bf16=self.emulate_bf16, emulate_fp8=self.book)
        loss = U.cross_entropy(logits, target_u8.long(), self.weight, self.ignore_index)
        prev, _FP8_BOOK = _FP8_BOOK, self.book       # (the data gradients of the fp8 layers quantise dy with the same book)
        try:
            grads = torch.autograd.grad(loss, [self.p[k] for k in self.trainable])
        finally:
            _FP8_BOOK = prev
        if self.book is not None:
            self.book.end_step()
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
                    newp, m, v = U.adam_step(p, g, m, v, self.nstep, lr, self.cfg["betas"], self.cfg["eps"],
                                             self.cfg["weight_decay"])
                    self.state[k] = (m, v)
                p.copy_(newp)
        return float(loss.detach()), logits.detach(), dict(zip(self.trainable, grads))

    def end_epoch(self):
        self.epoch += 1
