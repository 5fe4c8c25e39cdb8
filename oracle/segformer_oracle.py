"""CPU oracle for the reference's `SegformerMod` (/root/reference/source/scripts/nets.py:313-349, factory name at
/root/reference/source/scripts/utils.py:191-192) - TEST INFRASTRUCTURE ONLY.

The reference wraps `transformers.SegformerForSemanticSegmentation` (third-party; the reference pins no version, this image
carries transformers 5.15.0) and adds its own tail.  This file restates both in plain functional torch:

  x [B,3,S,S] raw 0..255
  -> preprocessor (nets.py:337-340): float32, (x - mean) / std with the ImageNet constants APPLIED TO THE RAW 0..255 VALUES
     (v2.ToDtype(float32) does not rescale)
  -> Mix Transformer encoder (Xie et al. 2021), 4 stages: overlapping patch embedding (conv k/s/p = 7/4/3, then 3/2/1) ->
     LayerNorm -> blocks [x + DropPath(Attn(LN(x))); x + DropPath(MixFFN(LN(x)))] -> LayerNorm.  Attention: queries from all
     tokens, keys / values from the map reduced by a conv k = s = sr_ratio + LayerNorm (sr_ratio 1: the tokens themselves),
     softmax(Q K^T / sqrt(d)) V per head, output projection.  Mix-FFN: fc1 -> depthwise 3x3 (bias) -> GELU(erf) -> fc2
  -> all-MLP decode head: per-stage Linear to `decoder_hidden_size`, bilinear (align_corners=False) to the stage-1 size,
     concat in REVERSE stage order, 1x1 conv (no bias) -> BatchNorm -> ReLU -> Dropout -> 1x1 classifier (NC, at S/4)
  -> the reference's tail `seq` (nets.py:331-335): ConvTranspose2d(NC,NC,8,2,3) -> ReLU -> ConvTranspose2d(NC,NC,4,2,1) -> ReLU
     -> Conv2d(NC,NC,3,padding=1): logits [B,NC,S,S]   (`self.upsampler` is constructed but never used)

`SegformerMod(classes)` with `pretrained=False` is `SegformerConfig()` = MiT-b0 (hidden 32/64/160/256, depths 2/2/2/2, heads
1/2/5/8, sr 8/4/2/1, decoder 256, drop-path 0.1, classifier dropout 0.1, LayerNorm eps 1e-6 in the CONFIG but the modules are built
as `nn.LayerNorm(hidden)` = eps 1e-5: restated as built); `pretrained=True` fetches MiT-b3 (64/128/320/512, depths 3/4/18/3,
decoder 768) - same code, `CONFIGS["b3"]`.

PINNED against the third-party implementation itself: tests/test_segformer_oracle_cpu.py builds
`SegformerForSemanticSegmentation(SegformerConfig())` from a LOCAL config (no fetch), wraps it exactly as nets.py:313-349 does,
loads the same weights and compares logits in eval mode, in train mode under the same torch RNG stream (`rng="torch"` makes the
same random calls in the same order: DropPath's `torch.rand((B,1,1))`, the head's `F.dropout`) and every parameter gradient.
Parameter names = the reference wrapper's state_dict keys under transformers 5.x (`segformer.segformer.stages.N...`); the 4.x names
a real CVCS checkpoint may carry (`segformer.segformer.encoder.block.N.M.attention.self.query...`) are mapped by `from_v4_name`.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.
"""
from __future__ import annotations

import math
import re
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import unet_oracle as U

CONFIGS = {
    "b0": dict(hidden=(32, 64, 160, 256), depths=(2, 2, 2, 2), heads=(1, 2, 5, 8), sr=(8, 4, 2, 1), patch=(7, 3, 3, 3), strides=(4, 2, 2, 2),
               mlp_ratio=4, decoder=256, drop_path=0.1, classifier_dropout=0.1, ln_eps=1e-5),
    "b3": dict(hidden=(64, 128, 320, 512), depths=(3, 4, 18, 3), heads=(1, 2, 5, 8), sr=(8, 4, 2, 1), patch=(7, 3, 3, 3), strides=(4, 2, 2, 2),
               mlp_ratio=4, decoder=768, drop_path=0.1, classifier_dropout=0.1, ln_eps=1e-5),
}
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)        # nets.py:339, applied to raw 0..255 values
BN_EPS, BN_MOMENTUM = 1e-5, 0.1
ENC, HEAD = "segformer.segformer.stages", "segformer.decode_head"


def drop_path_rates(cfg):
    """modeling_segformer.SegformerModel.__init__: linear decay over the blocks; rate 0 builds nn.Identity (no random call)"""
    n = sum(cfg["depths"])
    return [cfg["drop_path"] * i / max(n - 1, 1) for i in range(n)]


def param_spec(num_classes: int, variant="b0"):
    cfg = CONFIGS[variant]
    spec = []

    def lin(p, cin, cout):
        spec.extend([(p + ".weight", (cout, cin)), (p + ".bias", (cout,))])

    def ln(p, c):
        spec.extend([(p + ".weight", (c,)), (p + ".bias", (c,))])

    def conv(p, cin, cout, k, bias=True, groups=1):
        spec.append((p + ".weight", (cout, cin // groups, k, k)))
        if bias:
            spec.append((p + ".bias", (cout,)))

    cin = 3
    for s, (c, depth, sr, k) in enumerate(zip(cfg["hidden"], cfg["depths"], cfg["sr"], cfg["patch"])):
        conv(f"{ENC}.{s}.patch_embeddings.proj", cin, c, k)
        ln(f"{ENC}.{s}.patch_embeddings.layer_norm", c)
        for b in range(depth):
            p = f"{ENC}.{s}.blocks.{b}"
            ln(p + ".layernorm_before", c)
            for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
                lin(f"{p}.attention.{n}", c, c)
            if sr > 1:
                conv(p + ".attention.sequence_reduction.sequence_reduction", c, c, sr)
                ln(p + ".attention.sequence_reduction.layer_norm", c)
            ln(p + ".layernorm_after", c)
            lin(p + ".mlp.fc1", c, cfg["mlp_ratio"] * c)
            conv(p + ".mlp.dwconv.dwconv", cfg["mlp_ratio"] * c, cfg["mlp_ratio"] * c, 3, groups=cfg["mlp_ratio"] * c)
            lin(p + ".mlp.fc2", cfg["mlp_ratio"] * c, c)
        ln(f"{ENC}.{s}.layer_norm", c)
        cin = c
    D = cfg["decoder"]
    for s, c in enumerate(cfg["hidden"]):
        lin(f"{HEAD}.linear_projections.{s}.proj", c, D)
    conv(HEAD + ".linear_fuse", D * len(cfg["hidden"]), D, 1, bias=False)
    spec.extend([(HEAD + ".batch_norm.weight", (D,)), (HEAD + ".batch_norm.bias", (D,)), (HEAD + ".batch_norm.running_mean", (D,)),
                 (HEAD + ".batch_norm.running_var", (D,))])
    conv(HEAD + ".classifier", D, num_classes, 1)
    NC = num_classes
    spec.extend([("seq.0.weight", (NC, NC, 8, 8)), ("seq.0.bias", (NC,)), ("seq.2.weight", (NC, NC, 4, 4)), ("seq.2.bias", (NC,))])   # ConvTranspose2d: [in, out, k, k]
    conv("seq.4", NC, NC, 3)
    return spec


def is_buffer(name):
    return name.endswith("running_mean") or name.endswith("running_var")


_V4 = [
    (r"^segformer\.segformer\.encoder\.patch_embeddings\.(\d+)\.", r"segformer.segformer.stages.\1.patch_embeddings."),
    (r"^segformer\.segformer\.encoder\.layer_norm\.(\d+)\.", r"segformer.segformer.stages.\1.layer_norm."),
    (r"^segformer\.segformer\.encoder\.block\.(\d+)\.(\d+)\.", r"segformer.segformer.stages.\1.blocks.\2."),
    (r"\.layer_norm_1\.", ".layernorm_before."), (r"\.layer_norm_2\.", ".layernorm_after."),
    (r"\.attention\.self\.query\.", ".attention.q_proj."), (r"\.attention\.self\.key\.", ".attention.k_proj."),
    (r"\.attention\.self\.value\.", ".attention.v_proj."), (r"\.attention\.output\.dense\.", ".attention.o_proj."),
    (r"\.attention\.self\.sr\.", ".attention.sequence_reduction.sequence_reduction."),
    (r"\.attention\.self\.layer_norm\.", ".attention.sequence_reduction.layer_norm."),
    (r"\.mlp\.dense1\.", ".mlp.fc1."), (r"\.mlp\.dense2\.", ".mlp.fc2."),
    (r"^segformer\.decode_head\.linear_c\.(\d+)\.", r"segformer.decode_head.linear_projections.\1."),
]


def from_v4_name(k: str) -> str:
    """state_dict key of the reference wrapper under transformers 4.x -> the same tensor's key under 5.x (identity for 5.x keys)"""
    for pat, rep in _V4:
        k = re.sub(pat, rep, k)
    return k


def init_params(num_classes, seed=0, variant="b0"):
    """deterministic synthetic weights that keep activations O(1) through the depth (the library's N(0, 0.02) init leaves the
    branches near zero and would test little): fan-in scaled linear / conv weights, norm gammas in [0.5, 1.5], small biases"""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for name, shape in param_spec(num_classes, variant):
        if name.endswith("running_mean"):
            t = torch.zeros(shape)
        elif name.endswith("running_var"):
            t = torch.ones(shape)
        elif name.startswith("seq.") and len(shape) == 4:
            # ConvTranspose2d [in, out, k, k]: each output pixel sums (k/2)^2 * in products; Conv2d [out, in, 3, 3]: 9 * in
            fan = shape[0] * (shape[2] // 2) ** 2 if name != "seq.4.weight" else shape[1] * 9
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan)
        elif len(shape) == 4:
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[1] * shape[2] * shape[3]))
        elif len(shape) == 2:
            t = torch.randn(shape, generator=g) * math.sqrt(1.0 / shape[1])
        elif name.endswith(".weight"):                     # LayerNorm / BatchNorm gamma
            t = 0.5 + torch.rand(shape, generator=g)
        else:
            t = (torch.rand(shape, generator=g) - 0.5) * 0.2
        out[name] = t.float()
    return out


# --------------------------------------------------------------------------- bf16 emulation (as swin_upernet_oracle: round where the
# HIP bf16 path stores or feeds bf16 - weights of the linear / conv layers, every stored activation, the attention probabilities)
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


# --------------------------------------------------------------------------- pieces
def preprocess(x):
    """nets.py:337-340, 347"""
    mean = torch.tensor(MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).view(1, 3, 1, 1)
    return (x.to(torch.float32) - mean) / std


def attention(p, pre, x, H, W, heads, sr, eps):
    """modeling_segformer.SegformerAttention.forward + eager_attention_forward; x [B, N, C] (already LayerNormed)"""
    B, N, C = x.shape
    d = C // heads
    q = _q(F.linear(x, _q(p[pre + ".q_proj.weight"]), p[pre + ".q_proj.bias"])).view(B, N, heads, d).transpose(1, 2)
    kv = x
    if sr > 1:
        m = x.transpose(1, 2).reshape(B, C, H, W)
        m = F.conv2d(m, _q(p[pre + ".sequence_reduction.sequence_reduction.weight"]), p[pre + ".sequence_reduction.sequence_reduction.bias"], stride=sr)
        kv = _q(m).reshape(B, C, -1).transpose(1, 2)
        kv = _q(F.layer_norm(kv, (C,), p[pre + ".sequence_reduction.layer_norm.weight"], p[pre + ".sequence_reduction.layer_norm.bias"], eps))
    Nk = kv.shape[1]
    k = _q(F.linear(kv, _q(p[pre + ".k_proj.weight"]), p[pre + ".k_proj.bias"])).view(B, Nk, heads, d).transpose(1, 2)
    v = _q(F.linear(kv, _q(p[pre + ".v_proj.weight"]), p[pre + ".v_proj.bias"])).view(B, Nk, heads, d).transpose(1, 2)
    a = torch.matmul(q, k.transpose(2, 3)) * d ** -0.5
    a = _q(F.softmax(a, dim=-1, dtype=torch.float32))
    o = _q(torch.matmul(a, v)).transpose(1, 2).reshape(B, N, C)
    return _q(F.linear(o, _q(p[pre + ".o_proj.weight"]), p[pre + ".o_proj.bias"]))


def mix_ffn(p, pre, x, H, W):
    """modeling_segformer.SegformerMixMLP.forward (hidden_dropout_prob 0: the two Dropouts are identities)"""
    B, N, C = x.shape
    h = _q(F.linear(x, _q(p[pre + ".fc1.weight"]), p[pre + ".fc1.bias"]))
    Ch = h.shape[-1]
    m = h.transpose(1, 2).reshape(B, Ch, H, W)
    m = F.conv2d(m, p[pre + ".dwconv.dwconv.weight"], p[pre + ".dwconv.dwconv.bias"], padding=1, groups=Ch)
    h = _q(F.gelu(m.flatten(2).transpose(1, 2)))
    return _q(F.linear(h, _q(p[pre + ".fc2.weight"]), p[pre + ".fc2.bias"]))


class Stochastic:
    """where the train-mode random numbers come from.
    rng="torch": the library's own calls in the library's order (pins the oracle against transformers under one torch seed).
    masks=dict(drop_path=[[B] keep / keep_prob per random DropPath call], dropout=[B,D,H,W] keep / (1-p)): the values the HIP path
    drew (replayed on the host from its counter-based generator).  Neither: stochastic layers off (identity)."""

    def __init__(self, rng=None, masks=None):
        self.rng, self.masks, self.n = rng, masks, 0
        assert rng in (None, "torch")

    def drop_path(self, h, rate):
        if rate == 0.0:
            return h
        if self.masks is not None:
            m = self.masks["drop_path"][self.n].to(h.dtype).view(-1, 1, 1)
            self.n += 1
            return h * m
        if self.rng == "torch":       # SegformerDropPath.forward
            keep = 1 - rate
            r = torch.floor(torch.rand((h.shape[0], 1, 1), dtype=h.dtype) + keep)
            return h.div(keep) * r
        return h

    def dropout(self, h, p_drop):
        if p_drop == 0.0:
            return h
        if self.masks is not None:
            return h * self.masks["dropout"].to(h.dtype)
        if self.rng == "torch":
            return F.dropout(h, p_drop, training=True)
        return h


def encoder(p, x, cfg, st: Stochastic | None):
    """-> the four stage outputs [B, C_s, H_s, W_s]"""
    rates = drop_path_rates(cfg)
    feats, li = [], 0
    h = x
    for s, (C, depth, heads, sr, k, stride) in enumerate(zip(cfg["hidden"], cfg["depths"], cfg["heads"], cfg["sr"], cfg["patch"], cfg["strides"])):
        pre = f"{ENC}.{s}"
        w = p[pre + ".patch_embeddings.proj.weight"]
        h = F.conv2d(h, _q(w), p[pre + ".patch_embeddings.proj.bias"], stride=stride, padding=k // 2)
        B, _, H, W = h.shape
        t = _q(h).flatten(2).transpose(1, 2)
        t = _q(F.layer_norm(t, (C,), p[pre + ".patch_embeddings.layer_norm.weight"], p[pre + ".patch_embeddings.layer_norm.bias"], cfg["ln_eps"]))
        for b in range(depth):
            bp = f"{pre}.blocks.{b}"
            rate = rates[li] if st is not None else 0.0
            li += 1
            a = attention(p, bp + ".attention", _q(F.layer_norm(t, (C,), p[bp + ".layernorm_before.weight"], p[bp + ".layernorm_before.bias"], cfg["ln_eps"])),
                          H, W, heads, sr, cfg["ln_eps"])
            t = _q((st.drop_path(a, rate) if st is not None else a) + t)
            f = mix_ffn(p, bp + ".mlp", _q(F.layer_norm(t, (C,), p[bp + ".layernorm_after.weight"], p[bp + ".layernorm_after.bias"], cfg["ln_eps"])), H, W)
            t = _q((st.drop_path(f, rate) if st is not None else f) + t)
        t = _q(F.layer_norm(t, (C,), p[pre + ".layer_norm.weight"], p[pre + ".layer_norm.bias"], cfg["ln_eps"]))
        h = t.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
        feats.append(h)
    return feats


def decode_head(p, feats, cfg, train, st: Stochastic | None):
    """modeling_segformer.SegformerDecodeHead.forward; BatchNorm in train mode updates p's running statistics in place"""
    B = feats[0].shape[0]
    size = feats[0].shape[2:]
    ups = []
    for s, f in enumerate(feats):
        H, W = f.shape[2:]
        t = _q(F.linear(f.flatten(2).transpose(1, 2), _q(p[f"{HEAD}.linear_projections.{s}.proj.weight"]), p[f"{HEAD}.linear_projections.{s}.proj.bias"]))
        t = t.transpose(1, 2).reshape(B, -1, H, W)
        ups.append(_q(F.interpolate(t, size=size, mode="bilinear", align_corners=False)))
    h = F.conv2d(torch.cat(ups[::-1], dim=1), _q(p[HEAD + ".linear_fuse.weight"]))
    h = _q(h)
    rm, rv = p[HEAD + ".batch_norm.running_mean"], p[HEAD + ".batch_norm.running_var"]
    if train:
        with torch.no_grad():
            n = h.numel() / h.shape[1]
            mean = h.mean(dim=(0, 2, 3))
            var = h.var(dim=(0, 2, 3), unbiased=False)
            rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
            rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var * n / max(n - 1, 1))
        h = F.batch_norm(h, None, None, p[HEAD + ".batch_norm.weight"], p[HEAD + ".batch_norm.bias"], True, 0.0, BN_EPS)
    else:
        h = F.batch_norm(h, rm, rv, p[HEAD + ".batch_norm.weight"], p[HEAD + ".batch_norm.bias"], False, 0.0, BN_EPS)
    h = _q(F.relu(h))
    if train and st is not None:
        h = _q(st.dropout(h, cfg["classifier_dropout"]))
    return _q(F.conv2d(h, _q(p[HEAD + ".classifier.weight"]), p[HEAD + ".classifier.bias"]))


def tail(p, z):
    """nets.py:331-335, 349: the reference's own upsampling tail on the NC-channel logits"""
    z = _q(F.relu(F.conv_transpose2d(z, _q(p["seq.0.weight"]), p["seq.0.bias"], stride=2, padding=3)))
    z = _q(F.relu(F.conv_transpose2d(z, _q(p["seq.2.weight"]), p["seq.2.bias"], stride=2, padding=1)))
    return F.conv2d(z, _q(p["seq.4.weight"]), p["seq.4.bias"], padding=1)


def forward(p, x, train=False, variant="b0", rng=None, masks=None, return_parts=False, emulate_bf16=False):
    """x [B,3,S,S] raw 0..255 (S % 32 == 0) -> logits f32 [B,NC,S,S]   (nets.py:346-349)"""
    global _EMULATE
    cfg = CONFIGS[variant]
    st = Stochastic(rng, masks) if train else None
    prev, _EMULATE = _EMULATE, bool(emulate_bf16)
    try:
        feats = encoder(p, _q(preprocess(x)), cfg, st)
        z = decode_head(p, feats, cfg, train, st)
        logits = tail(p, z)
    finally:
        _EMULATE = prev
    return (logits, feats, z) if return_parts else logits


def forward_gflop_per_tile(num_classes, S, variant="b0"):
    """algorithmic forward GFLOP per tile (2 x multiply-adds of every linear / conv layer and of QK^T, PV)"""
    cfg = CONFIGS[variant]
    fl, cin, H = 0.0, 3, S
    for C, depth, heads, sr, k, stride in zip(cfg["hidden"], cfg["depths"], cfg["heads"], cfg["sr"], cfg["patch"], cfg["strides"]):
        H //= stride
        N = H * H
        Nk = N // (sr * sr)
        fl += 2.0 * N * C * cin * k * k
        per = 2.0 * N * C * C * 2 + 2.0 * Nk * C * C * 2 + (2.0 * Nk * C * C * sr * sr if sr > 1 else 0.0)     # q, o; k, v; reduction conv
        per += 2.0 * 2 * N * Nk * C                                                                           # QK^T, PV
        Ch = cfg["mlp_ratio"] * C
        per += 2.0 * N * C * Ch * 2 + 2.0 * N * Ch * 9
        fl += depth * per
        cin = C
    D, h1 = cfg["decoder"], S // 4
    Hs = [S // 4, S // 8, S // 16, S // 32]
    fl += sum(2.0 * hs * hs * c * D for hs, c in zip(Hs, cfg["hidden"]))
    fl += 2.0 * h1 * h1 * 4 * D * D + 2.0 * h1 * h1 * D * num_classes
    NC = num_classes
    fl += 2.0 * (2 * h1) ** 2 * NC * NC * 16 + 2.0 * (4 * h1) ** 2 * NC * NC * 4 + 2.0 * S * S * NC * NC * 9
    return fl / 1e9


class OracleTrainer:
    """forward -> loss -> zero_grad -> backward -> opt.step (S/train.py:121-126) on the definition above"""

    def __init__(self, num_classes, opt="SGD2", epochs=20, ignore_index=-100, weight=None, seed=0, params=None, variant="b0", emulate_bf16=False):
        self.variant, self.emulate_bf16 = variant, emulate_bf16
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

    def step(self, x_u8, target_u8, masks=None, rng=None):
        logits = forward(self.p, x_u8.to(torch.float32), train=True, variant=self.variant, masks=masks, rng=rng, emulate_bf16=self.emulate_bf16)
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
