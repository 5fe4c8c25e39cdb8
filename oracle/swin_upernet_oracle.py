"""CPU oracle for Swin-T + UPerNet (BASELINE.json configs[3]: "Swin-T encoder + UPerNet decoder, 512x512 window-7 attention";
the reference documents `TSwin` / `BSwin` at configs/train/README.txt:44-45 but ships no definition) - TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: no reference code exists for this model.  This file restates the PUBLISHED architectures in plain torch -
Swin Transformer (Liu et al. 2021) as a backbone and the UPerNet head (Xiao et al. 2018) - with exactly the structure and
parameter names of `transformers.UperNetForSemanticSegmentation` over a `SwinBackbone` (transformers 5.15.0 in this image,
built from a LOCAL config, no fetch: SURVEY.md section 8c); `tests/test_swin_oracle_cpu.py` loads the same weights into that
independent implementation and compares logits and every stage output.  Configuration: embed_dim 96, depths (2,2,6,2), heads
(3,6,12,24), window 7, mlp ratio 4, qkv bias, LayerNorm eps 1e-5, GELU (erf), drop-path / dropout 0 (stochastic layers off:
they cannot be checked and do not change the work), no auxiliary head; UPerNet hidden size 512, pool scales (1,2,3,6).

  x [B,3,S,S] raw 0..255 (S % 32 == 0)
  -> patch embedding: conv 4x4 / stride 4 (3 -> 96, bias) -> LayerNorm                                   tokens [B, S/4 * S/4, 96]
  -> 4 stages of Swin blocks: x + W-MSA/SW-MSA(LN(x)); x + MLP(LN(x)).  Windows of 7x7 tokens on the map zero-padded (AFTER the
     LayerNorm) to a multiple of 7, cyclic shift 3 in every second block, relative-position bias (169 x heads table), region
     mask (-100) for shifted windows; patch merging (2x2 neighbours -> LayerNorm(4C) -> Linear 4C -> 2C) between stages
  -> per-stage output LayerNorm -> UPerNet: PPM (adaptive average pools 1/2/3/6 -> 1x1 conv+BN+ReLU -> bilinear resize) + 3x3
     bottleneck on the last stage, 1x1 laterals, top-down adds, 3x3 FPN convs, all resized to stride 4, concat, 3x3 fusion conv,
     1x1 classifier -> bilinear x4 to the input size.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import unet_oracle as U

EMBED, DEPTHS, HEADS, WINDOW, MLP_RATIO, LN_EPS = 96, (2, 2, 6, 2), (3, 6, 12, 24), 7, 4, 1e-5
HIDDEN, POOL_SCALES = 512, (1, 2, 3, 6)
# (embed dim, depths, heads): Swin-T (`TSwin`) and Swin-B (`BSwin`, configs/train/README.txt:44-45); head dim 32 in both
VARIANTS = {"tiny": (96, (2, 2, 6, 2), (3, 6, 12, 24)), "base": (128, (2, 2, 18, 2), (4, 8, 16, 32))}


def stage_dims(variant="tiny"):
    return [VARIANTS[variant][0] * 2 ** i for i in range(4)]


def param_spec(num_classes: int, variant="tiny"):
    EMBED, DEPTHS, HEADS = VARIANTS[variant]      # noqa: N806  (shadow the Swin-T module constants)
    stage_dims = lambda: [EMBED * 2 ** i for i in range(4)]   # noqa: E731
    spec = []

    def lin(p, cin, cout, bias=True):
        spec.append((p + ".weight", (cout, cin)))
        if bias:
            spec.append((p + ".bias", (cout,)))

    def ln(p, c):
        spec.extend([(p + ".weight", (c,)), (p + ".bias", (c,))])

    def cbr(p, cin, cout, k):
        spec.append((p + ".conv.weight", (cout, cin, k, k)))
        spec.extend([(p + ".batch_norm.weight", (cout,)), (p + ".batch_norm.bias", (cout,)),
                     (p + ".batch_norm.running_mean", (cout,)), (p + ".batch_norm.running_var", (cout,))])

    spec.append(("backbone.swin.embeddings.patch_embeddings.projection.weight", (EMBED, 3, 4, 4)))
    spec.append(("backbone.swin.embeddings.patch_embeddings.projection.bias", (EMBED,)))
    ln("backbone.swin.embeddings.norm", EMBED)
    for s, (depth, dim, heads) in enumerate(zip(DEPTHS, stage_dims(), HEADS)):
        for b in range(depth):
            p = f"backbone.swin.encoder.layers.{s}.blocks.{b}"
            for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
                lin(f"{p}.attention.{n}", dim, dim)
            spec.append((f"{p}.attention.relative_position_bias.relative_position_bias_table", ((2 * WINDOW - 1) ** 2, heads)))
            ln(p + ".layernorm_before", dim)
            ln(p + ".layernorm_after", dim)
            lin(p + ".mlp.fc1", dim, MLP_RATIO * dim)
            lin(p + ".mlp.fc2", MLP_RATIO * dim, dim)
        if s < 3:
            lin(f"backbone.swin.encoder.layers.{s}.downsample.reduction", 4 * dim, 2 * dim, bias=False)
            ln(f"backbone.swin.encoder.layers.{s}.downsample.norm", 4 * dim)
    for s, dim in enumerate(stage_dims(), start=1):
        ln(f"backbone.hidden_states_norms.stage{s}", dim)
    spec.append(("decode_head.classifier.weight", (num_classes, HIDDEN, 1, 1)))
    spec.append(("decode_head.classifier.bias", (num_classes,)))
    for i in range(len(POOL_SCALES)):
        cbr(f"decode_head.psp_modules.{i}.1", stage_dims()[3], HIDDEN, 1)
    cbr("decode_head.bottleneck", stage_dims()[3] + len(POOL_SCALES) * HIDDEN, HIDDEN, 3)
    for i in range(3):
        cbr(f"decode_head.lateral_convs.{i}", stage_dims()[i], HIDDEN, 1)
    for i in range(3):
        cbr(f"decode_head.fpn_convs.{i}", HIDDEN, HIDDEN, 3)
    cbr("decode_head.fpn_bottleneck", 4 * HIDDEN, HIDDEN, 3)
    return spec


def is_buffer(name):
    return name.endswith("running_mean") or name.endswith("running_var")


def init_params(num_classes, seed=0, variant="tiny"):
    """deterministic synthetic weights: linear / conv ~ N(0, 1/fan_in)-ish scales that keep activations O(1), LayerNorm / BatchNorm
    gamma in [0.5, 1.5], small biases, relative-position tables ~ N(0, 0.5)"""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for name, shape in param_spec(num_classes, variant):
        if name.endswith("running_mean"):
            t = torch.zeros(shape)
        elif name.endswith("running_var"):
            t = torch.ones(shape)
        elif name.endswith("relative_position_bias_table"):
            t = torch.randn(shape, generator=g) * 0.5
        elif len(shape) == 4:
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[1] * shape[2] * shape[3]))
        elif len(shape) == 2:
            t = torch.randn(shape, generator=g) * math.sqrt(1.0 / shape[1])
        elif name.endswith("norm.weight") or name.endswith("layernorm_before.weight") or name.endswith("layernorm_after.weight") \
                or ("hidden_states_norms" in name and name.endswith(".weight")) or name.endswith("batch_norm.weight"):
            t = 0.5 + torch.rand(shape, generator=g)
        else:
            t = (torch.rand(shape, generator=g) - 0.5) * 0.2
        out[name] = t.float()
    # the patch embedding sees raw 0..255 pixels: scale its weights so that the tokens are O(1) before the LayerNorm anyway
    return out


# --------------------------------------------------------------------------- bf16 emulation
# forward(..., emulate_bf16=True) rounds to bfloat16 where the HIP bf16 path stores or feeds bf16: the weights of every linear / conv
# layer but the classifier, every stored activation (linear / LayerNorm / GELU / attention / residual-add / BatchNorm+ReLU / pooling /
# resize outputs) and the attention probabilities that enter P.V; sums, softmax, LayerNorm and BatchNorm arithmetic stay f32.
_EMULATE = False


class _RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def _q(x):
    return _RoundSTE.apply(x) if _EMULATE else x


# --------------------------------------------------------------------------- Swin pieces
def relative_position_index():
    c = torch.stack(torch.meshgrid(torch.arange(WINDOW), torch.arange(WINDOW), indexing="ij")).flatten(1)   # 2, 49
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += WINDOW - 1
    rel[:, :, 1] += WINDOW - 1
    rel[:, :, 0] *= 2 * WINDOW - 1
    return rel.sum(-1)   # [49, 49]


def window_partition(x):
    B, H, W, C = x.shape
    x = x.view(B, H // WINDOW, WINDOW, W // WINDOW, WINDOW, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, WINDOW * WINDOW, C)


def window_reverse(w, H, W):
    C = w.shape[-1]
    x = w.view(-1, H // WINDOW, W // WINDOW, WINDOW, WINDOW, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, H, W, C)


def shift_mask(Hp, Wp, shift, dtype):
    """[nW, 49, 49]: 0 inside a cyclic-shift region, -100 across regions (Swin's SW-MSA mask)"""
    h = torch.arange(Hp)
    w = torch.arange(Wp)
    hr = (h >= Hp - WINDOW).long() + (h >= Hp - shift).long()
    wr = (w >= Wp - WINDOW).long() + (w >= Wp - shift).long()
    img = (hr[None, :, None, None] * 3 + wr[None, None, :, None]).to(dtype)
    mw = window_partition(img).view(-1, WINDOW * WINDOW)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def swin_block(x, H, W, p, pre, heads, shift):
    """x [B, H*W, C]"""
    B, L, C = x.shape
    h = _q(F.layer_norm(x, (C,), p[pre + ".layernorm_before.weight"], p[pre + ".layernorm_before.bias"], LN_EPS)).view(B, H, W, C)
    pr, pb = (WINDOW - W % WINDOW) % WINDOW, (WINDOW - H % WINDOW) % WINDOW
    h = F.pad(h, (0, 0, 0, pr, 0, pb))                       # zero tokens AFTER the LayerNorm
    Hp, Wp = H + pb, W + pr
    if shift:
        h = torch.roll(h, shifts=(-shift, -shift), dims=(1, 2))
    win = window_partition(h)                                # [B*nW, 49, C]
    d = C // heads
    q, k, v = (_q(F.linear(win, _q(p[f"{pre}.attention.{n}.weight"]), p[f"{pre}.attention.{n}.bias"])).view(-1, 49, heads, d).transpose(1, 2)
               for n in ("q_proj", "k_proj", "v_proj"))
    bias = p[pre + ".attention.relative_position_bias.relative_position_bias_table"][relative_position_index().view(-1)]
    bias = bias.view(49, 49, heads).permute(2, 0, 1).unsqueeze(0)            # [1, heads, 49, 49]
    att = (q @ k.transpose(-1, -2)) * d ** -0.5 + bias
    if shift:
        m = shift_mask(Hp, Wp, shift, att.dtype)                             # [nW, 49, 49]
        att = (att.view(B, -1, heads, 49, 49) + m[None, :, None]).view(-1, heads, 49, 49)
    att = _q(torch.softmax(att, dim=-1))
    o = _q((att @ v).transpose(1, 2).reshape(-1, 49, C))
    o = _q(F.linear(o, _q(p[pre + ".attention.o_proj.weight"]), p[pre + ".attention.o_proj.bias"]))
    o = window_reverse(o, Hp, Wp)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    o = o[:, :H, :W, :].reshape(B, L, C)
    x = _q(x + o)
    h = _q(F.layer_norm(x, (C,), p[pre + ".layernorm_after.weight"], p[pre + ".layernorm_after.bias"], LN_EPS))
    h = _q(F.gelu(_q(F.linear(h, _q(p[pre + ".mlp.fc1.weight"]), p[pre + ".mlp.fc1.bias"]))))
    return _q(x + _q(F.linear(h, _q(p[pre + ".mlp.fc2.weight"]), p[pre + ".mlp.fc2.bias"])))


def patch_merge(x, H, W, p, pre):
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    x = torch.cat([x[:, r::2, c::2, :] for c in range(2) for r in range(2)], dim=-1).view(B, -1, 4 * C)
    x = _q(F.layer_norm(x, (4 * C,), p[pre + ".norm.weight"], p[pre + ".norm.bias"], LN_EPS))
    return _q(F.linear(x, _q(p[pre + ".reduction.weight"])))


def backbone(p, x):
    """-> [stage1 .. stage4] feature maps NCHW after the per-stage output LayerNorm (what UPerNet consumes)"""
    y = _q(F.conv2d(x, _q(p["backbone.swin.embeddings.patch_embeddings.projection.weight"]),
                    p["backbone.swin.embeddings.patch_embeddings.projection.bias"], stride=4))
    B, C, H, W = y.shape
    t = y.flatten(2).transpose(1, 2)
    t = _q(F.layer_norm(t, (C,), p["backbone.swin.embeddings.norm.weight"], p["backbone.swin.embeddings.norm.bias"], LN_EPS))
    feats = []
    depths = [1 + max(int(k.split(".")[6]) for k in p if k.startswith(f"backbone.swin.encoder.layers.{s}.blocks.")) for s in range(4)]
    heads_ = [p[f"backbone.swin.encoder.layers.{s}.blocks.0.attention.relative_position_bias.relative_position_bias_table"].shape[1] for s in range(4)]
    for s, (depth, heads) in enumerate(zip(depths, heads_)):     # (Swin-T or Swin-B: read off the parameters)
        for b in range(depth):
            t = swin_block(t, H, W, p, f"backbone.swin.encoder.layers.{s}.blocks.{b}", heads, 0 if b % 2 == 0 else WINDOW // 2)
        C = t.shape[-1]
        n = _q(F.layer_norm(t, (C,), p[f"backbone.hidden_states_norms.stage{s + 1}.weight"], p[f"backbone.hidden_states_norms.stage{s + 1}.bias"], LN_EPS))
        feats.append(n.view(B, H, W, C).permute(0, 3, 1, 2).contiguous())
        if s < 3:
            t = patch_merge(t, H, W, p, f"backbone.swin.encoder.layers.{s}.downsample")
            H, W = H // 2, W // 2
    return feats


# --------------------------------------------------------------------------- UPerNet head
def _cbr(x, p, pre, pad, train):
    y = _q(F.conv2d(x, _q(p[pre + ".conv.weight"]), None, padding=pad))
    y = F.batch_norm(y, p[pre + ".batch_norm.running_mean"], p[pre + ".batch_norm.running_var"], p[pre + ".batch_norm.weight"],
                     p[pre + ".batch_norm.bias"], training=train, momentum=U.BN_MOMENTUM, eps=U.BN_EPS)
    return _q(F.relu(y))


def upernet_head(p, feats, train):
    x = feats[-1]
    outs = [x]
    for i, s in enumerate(POOL_SCALES):
        o = _cbr(_q(F.adaptive_avg_pool2d(x, s)), p, f"decode_head.psp_modules.{i}.1", 0, train)
        outs.append(_q(F.interpolate(o, size=x.shape[2:], mode="bilinear", align_corners=False)))
    laterals = [_cbr(feats[i], p, f"decode_head.lateral_convs.{i}", 0, train) for i in range(3)]
    laterals.append(_cbr(torch.cat(outs, 1), p, "decode_head.bottleneck", 1, train))
    for i in range(3, 0, -1):
        laterals[i - 1] = _q(laterals[i - 1] + _q(F.interpolate(laterals[i], size=laterals[i - 1].shape[2:], mode="bilinear", align_corners=False)))
    fpn = [_cbr(laterals[i], p, f"decode_head.fpn_convs.{i}", 1, train) for i in range(3)] + [laterals[3]]
    for i in range(3, 0, -1):
        fpn[i] = _q(F.interpolate(fpn[i], size=fpn[0].shape[2:], mode="bilinear", align_corners=False))
    y = _cbr(torch.cat(fpn, 1), p, "decode_head.fpn_bottleneck", 1, train)
    return F.conv2d(y, p["decode_head.classifier.weight"], p["decode_head.classifier.bias"])


def forward(p, x, train=False, return_feats=False, emulate_bf16=False):
    """x f32 [B,3,S,S] raw 0..255 -> logits f32 [B,NC,S,S]"""
    global _EMULATE
    prev, _EMULATE = _EMULATE, bool(emulate_bf16)
    try:
        feats = backbone(p, x)
        z = upernet_head(p, feats, train)
        logits = F.interpolate(z, size=x.shape[2:], mode="bilinear", align_corners=False)
    finally:
        _EMULATE = prev
    return (logits, feats) if return_feats else logits


def forward_gflop_per_tile(num_classes, S, variant="tiny"):
    """algorithmic forward GFLOP per tile: linear layers, attention (QK^T and AV over the padded windows), convolutions"""
    EMBED, DEPTHS, HEADS = VARIANTS[variant]      # noqa: N806
    fl = 2.0 * 3 * 16 * EMBED * (S // 4) ** 2
    H = S // 4
    for s, (depth, dim, heads) in enumerate(zip(DEPTHS, stage_dims(variant), HEADS)):
        L = H * H
        Hp = -(-H // WINDOW) * WINDOW
        Lp = Hp * Hp
        per_block = 2.0 * Lp * dim * dim * 4 + 2.0 * 2 * Lp * 49 * dim + 2.0 * L * dim * MLP_RATIO * dim * 2
        fl += depth * per_block
        if s < 3:
            fl += 2.0 * (L // 4) * 4 * dim * 2 * dim
            H //= 2
    d = stage_dims(variant)
    h4, h1 = S // 32, S // 4
    fl += sum(2.0 * d[3] * HIDDEN * s * s for s in POOL_SCALES)
    fl += 2.0 * (d[3] + 4 * HIDDEN) * HIDDEN * 9 * h4 * h4
    for i in range(3):
        hi = S // (4 * 2 ** i)
        fl += 2.0 * d[i] * HIDDEN * hi * hi + 2.0 * HIDDEN * HIDDEN * 9 * hi * hi
    fl += 2.0 * 4 * HIDDEN * HIDDEN * 9 * h1 * h1 + 2.0 * HIDDEN * num_classes * h1 * h1
    return fl / 1e9


class OracleTrainer:
    """forward -> loss -> zero_grad -> backward -> opt.step (S/train.py:121-126) on the definition above"""

    def __init__(self, num_classes, opt="SGD2", epochs=20, ignore_index=-100, weight=None, seed=0, params=None, emulate_bf16=False,
                 variant="tiny"):
        self.emulate_bf16 = emulate_bf16
        self.p = params if params is not None else init_params(num_classes, seed, variant)
        self.trainable = [k for k in self.p if not is_buffer(k)]
        for k in self.trainable:
            self.p[k].requires_grad_(True)
        self.cfg = dict(U.OPTIMIZERS[opt])
        if self.cfg["total_iters"] is None:
            self.cfg["total_iters"] = epochs
        self.state, self.nstep, self.epoch = {}, 0, 0
        self.ignore_index, self.weight = ignore_index, weight

    def lr(self):
        return U.polynomial_lr(self.cfg["lr"], self.epoch, self.cfg["total_iters"], self.cfg["power"])

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
