"""CPU oracle for the per-tile segmentation hot path (TEST INFRASTRUCTURE ONLY).

This file is a torch-CPU / fp32 restatement of the reference's algorithm for the
path SURVEY.md section 8 names.  It is the checker for the HIP kernels; it is never
imported by the product package `cvcs_amd` (only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it).

Pinning: the reference (theElandor/CVCS) ships no tests, golden vectors or
fixtures.  The oracle is pinned by outputs of the reference's own Python run in the
build container (`oracle/make_golden.py` imports `/root/reference/source/scripts`
`blocks.py` directly and `nets.py` through an in-process torchvision placeholder) and
committed as `tests/golden/*.npz`; `tests/test_oracle_golden.py` replays them.

All citations are `path:line` into `/root/reference/source/scripts/` (`S/`).
The arithmetic of the reference lives in torch (README.MD:19 pins torch==2.3.1;
this image has 2.10.0+rocm7.0); the oracle calls the same ATen CPU ops in fp32, plus
`numpy_*` twins for the small ops so that a torch version drift shows up.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

WIDTHS = (64, 128, 256, 512, 1024)  # S/nets.py:126-147
BN_EPS = 1e-5  # nn.BatchNorm2d default, S/blocks.py:14
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------- params
def param_spec(variant: str, num_classes: int):
    """Ordered (name, shape) list == `nets.Urnet/Urnetv2(num_classes).state_dict()` minus
    `num_batches_tracked` (S/nets.py:34-199, S/blocks.py:8-49).  Order is the
    module registration order, which is also `net.parameters()` order for the
    non-buffer entries."""
    assert variant in ("Unet", "Unetv2")
    spec = []

    def conv(prefix, cin, cout, k=3):
        spec.append((prefix + ".weight", (cout, cin, k, k)))
        spec.append((prefix + ".bias", (cout,)))

    def bn(prefix, c):
        spec.append((prefix + ".weight", (c,)))
        spec.append((prefix + ".bias", (c,)))
        spec.append((prefix + ".running_mean", (c,)))
        spec.append((prefix + ".running_var", (c,)))

    cin = 3
    for lvl, w in enumerate(WIDTHS, start=1):
        base = 0 if lvl == 1 else 1  # encode2..5 start with MaxPool2d at index 0 (S/nets.py:129-147)
        for j in range(2):
            p = f"encode{lvl}.{base + j}.layer"
            conv(p + ".0", cin if j == 0 else w, w)
            bn(p + ".1", w)
        cin = w
    for d in range(1, 5):
        cin = WIDTHS[5 - d]
        cout = cin // 2
        if variant == "Unetv2":  # nn.ConvTranspose2d(cin, cout, 2, 2): weight [cin, cout, 2, 2] (S/nets.py:150)
            spec.append((f"upscale{d}.0.weight", (cin, cout, 2, 2)))
            spec.append((f"upscale{d}.0.bias", (cout,)))
        else:  # UnetUpscaleLayer: Upsample at layer.0, conv3x3 at layer.1 (S/blocks.py:25-34)
            conv(f"upscale{d}.0.layer.1", cin, cout)
        p = f"decode_forward{d}.0.layer"
        conv(p + ".0", cin, cout)
        bn(p + ".2", cout)
        conv(p + ".3", cout, cout)
        bn(p + ".5", cout)
    conv("decode_forward4.1", 64, num_classes, k=1)  # S/nets.py:172
    return spec


def is_buffer(name: str) -> bool:
    return name.endswith("running_mean") or name.endswith("running_var")


def init_params(variant: str, num_classes: int, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic synthetic weights (NOT torch's default init: independent of torch
    version).  Conv weights ~ N(0, 2/fan_in) (keeps activations O(1) through 23 layers),
    BN gamma in [0.5,1.5], beta small, running stats at their defaults (0, 1)."""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for name, shape in param_spec(variant, num_classes):
        if name.endswith("running_mean"):
            t = torch.zeros(shape)
        elif name.endswith("running_var"):
            t = torch.ones(shape)
        elif len(shape) == 4:
            if name.startswith("upscale") and variant == "Unetv2":
                fan_in = shape[0]  # each output pixel sees exactly one tap
            else:
                fan_in = shape[1] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        elif (name.startswith("encode") and name.endswith(".layer.1.weight")) or \
                name.endswith(".layer.2.weight") or name.endswith(".layer.5.weight"):
            t = 0.5 + torch.rand(shape, generator=g)
        else:
            t = (torch.rand(shape, generator=g) - 0.5) * 0.2
        out[name] = t.float()
    return out


# --------------------------------------------------------------------------- blocks
def batch_norm(x, p, prefix, train: bool, update_running: bool = True):
    """nn.BatchNorm2d defaults (S/blocks.py:14,42,45): train -> batch mean / biased var
    for normalisation, running stats updated with the UNBIASED var, momentum 0.1."""
    rm, rv = p[prefix + ".running_mean"], p[prefix + ".running_var"]
    if train and not update_running:
        rm, rv = rm.clone(), rv.clone()
    return F.batch_norm(x, rm, rv, p[prefix + ".weight"], p[prefix + ".bias"],
                        training=train, momentum=BN_MOMENTUM, eps=BN_EPS)


def encode_layer(x, p, prefix, train):
    """UnetEncodeLayer.forward: ReLU(BN(conv3x3(x)+b)), padding=1 (S/blocks.py:8-23)."""
    y = F.conv2d(x, p[prefix + ".0.weight"], p[prefix + ".0.bias"], padding=1)
    return F.relu(batch_norm(y, p, prefix + ".1", train))


def decode_layer(x, p, prefix, train):
    """UnetForwardDecodeLayer.forward: (conv3x3 -> ReLU -> BN) twice - ReLU BEFORE BN
    (S/blocks.py:36-49)."""
    y = F.relu(F.conv2d(x, p[prefix + ".0.weight"], p[prefix + ".0.bias"], padding=1))
    y = batch_norm(y, p, prefix + ".2", train)
    y = F.relu(F.conv2d(y, p[prefix + ".3.weight"], p[prefix + ".3.bias"], padding=1))
    return batch_norm(y, p, prefix + ".5", train)


def upscale(x, p, d, variant):
    if variant == "Unetv2":  # S/nets.py:150,156,162,168
        return F.conv_transpose2d(x, p[f"upscale{d}.0.weight"], p[f"upscale{d}.0.bias"], stride=2)
    # UnetUpscaleLayer: nn.Upsample(scale 2, bilinear) (align_corners=False) then conv3x3 (S/blocks.py:25-34)
    y = F.interpolate(x, scale_factor=(2, 2), mode="bilinear")
    return F.conv2d(y, p[f"upscale{d}.0.layer.1.weight"], p[f"upscale{d}.0.layer.1.bias"], padding=1)


def center_crop(x, size):
    """torchvision.transforms.functional.center_crop for the only call site S/nets.py:109,193
    (identity whenever H = W is a multiple of 16, which every supported patch size is)."""
    h, w = x.shape[-2:]
    if h == size and w == size:
        return x
    top = int(round((h - size) / 2.0))
    left = int(round((w - size) / 2.0))
    return x[..., top:top + size, left:left + size]


def unet_forward(p, x, variant="Unetv2", train=False, return_acts=False):
    """nets.Urnet.forward (S/nets.py:93-115) / nets.Urnetv2.forward (S/nets.py:176-199).
    x: f32 [B,3,S,S] raw 0..255 values (S/train.py:121; no normalisation)."""
    acts = OrderedDict()
    xs = []
    h = x
    for lvl in range(1, 6):
        if lvl > 1:
            h = F.max_pool2d(h, kernel_size=2, stride=2)  # S/nets.py:130,135,140,145
        base = 0 if lvl == 1 else 1
        h = encode_layer(h, p, f"encode{lvl}.{base}.layer", train)
        h = encode_layer(h, p, f"encode{lvl}.{base + 1}.layer", train)
        xs.append(h)
        acts[f"x{lvl}"] = h
    x1, x2, x3, x4, x5 = xs
    y = upscale(x5, p, 1, variant)
    y = decode_layer(torch.cat((x4, y), 1), p, "decode_forward1.0.layer", train)  # S/nets.py:185
    y = upscale(y, p, 2, variant)
    y = decode_layer(torch.cat((x3, y), 1), p, "decode_forward2.0.layer", train)  # S/nets.py:189
    y = upscale(y, p, 3, variant)
    # level 3 concatenates (up, skip) - the other levels (skip, up)  (S/nets.py:193)
    y = decode_layer(torch.cat((center_crop(y, x2.shape[2]), x2), 1), p, "decode_forward3.0.layer", train)
    y = upscale(y, p, 4, variant)
    y = decode_layer(torch.cat((x1, y), 1), p, "decode_forward4.0.layer", train)  # S/nets.py:197
    acts["d4"] = y
    logits = F.conv2d(y, p["decode_forward4.1.weight"], p["decode_forward4.1.bias"])  # S/nets.py:172
    return (logits, acts) if return_acts else logits


# --------------------------------------------------------------------------- loss
def cross_entropy(logits, target, weight=None, ignore_index=-100):
    """nn.CrossEntropyLoss(weight, ignore_index) as built at S/utils.py:230,238 and called at
    S/train.py:122: sum_{t!=ignore} -w[t] log_softmax(z)[t] / sum_{t!=ignore} w[t]."""
    return F.cross_entropy(logits, target.long(), weight=weight, ignore_index=ignore_index)


def numpy_cross_entropy(logits, target, weight=None, ignore_index=-100):
    """Independent float64 numpy twin of `cross_entropy`; returns (loss, dlogits)."""
    z = np.asarray(logits, dtype=np.float64)
    t = np.asarray(target).astype(np.int64)
    B, C = z.shape[:2]
    zz = np.moveaxis(z, 1, -1).reshape(-1, C)
    tt = t.reshape(-1)
    w = np.ones(C) if weight is None else np.asarray(weight, dtype=np.float64)
    m = zz.max(1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(zz - m).sum(1))
    valid = tt != ignore_index
    ts = np.where(valid, tt, 0)
    wt = np.where(valid, w[ts], 0.0)
    den = wt.sum()
    loss = (wt * (lse - zz[np.arange(len(tt)), ts])).sum() / den
    sm = np.exp(zz - lse[:, None])
    sm[np.arange(len(tt)), ts] -= 1.0
    d = sm * (wt / den)[:, None]
    d = np.moveaxis(d.reshape(z.shape[0], *z.shape[2:], C), -1, 1)
    return loss, d


def class_weights(counts, ignore_background=False):
    """Loader.get_class_weights (S/dataset.py:360-384): w_j = N_tot / (bins * N_j), 0 for empty
    classes; with ignore_background the background count is dropped from N_tot and bins and
    its weight is 0."""
    counts = np.asarray(counts, dtype=np.float32)
    c = counts[1:] if ignore_background else counts
    tot = np.float32(c.sum())
    bins = len(c)
    w = np.array([0.0 if n == 0 else tot / (bins * n) for n in c], dtype=np.float32)
    if ignore_background:
        w = np.concatenate(([np.float32(0)], w))
    return w


# --------------------------------------------------------------------------- optimiser
OPTIMIZERS = {
    # S/utils.py:213-218 (SGD1 is unreachable in the reference: quirk 2 of SURVEY section 5)
    "SGD1": dict(kind="sgd", lr=0.001, momentum=0.9, weight_decay=1e-5, total_iters=5, power=1.0),
    "SGD2": dict(kind="sgd", lr=0.006, momentum=0.9, weight_decay=1e-5, total_iters=20, power=1.0),
    "ADAM1": dict(kind="adam", lr=0.005, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                  total_iters=None, power=2.0),  # total_iters = config['epochs']
}


def polynomial_lr(base_lr, epoch, total_iters, power):
    """Closed form of torch.optim.lr_scheduler.PolynomialLR after `epoch` scheduler.step()
    calls (S/utils.py:212,215,218; stepped once per epoch S/train.py:132-133)."""
    e = min(epoch, total_iters)
    return base_lr * (1.0 - e / total_iters) ** power


def sgd_step(p, g, buf, lr, momentum, weight_decay, first):
    """torch.optim.SGD (dampening 0, nesterov False): g += wd*p; buf = g (first) or
    momentum*buf + g; p -= lr*buf."""
    g = g + weight_decay * p
    buf = g.clone() if first else momentum * buf + g
    return p - lr * buf, buf


def adam_step(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam (amsgrad False), `step` is 1-based."""
    if weight_decay:
        g = g + weight_decay * p
    m = betas[0] * m + (1 - betas[0]) * g
    v = betas[1] * v + (1 - betas[1]) * g * g
    bc1 = 1 - betas[0] ** step
    bc2 = 1 - betas[1] ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * (m / denom), m, v


class OracleTrainer:
    """One reference training step, in the order of S/train.py:121-126:
    forward -> loss -> zero_grad -> backward -> opt.step."""

    def __init__(self, variant, num_classes, opt="SGD2", epochs=20, ignore_index=-100,
                 weight=None, seed=0, params=None):
        self.variant = variant
        self.p = params if params is not None else init_params(variant, num_classes, seed)
        self.trainable = [k for k in self.p if not is_buffer(k)]
        for k in self.trainable:
            self.p[k].requires_grad_(True)
        self.cfg = dict(OPTIMIZERS[opt])
        if self.cfg["total_iters"] is None:
            self.cfg["total_iters"] = epochs
        self.state = {}
        self.nstep = 0
        self.epoch = 0
        self.ignore_index = ignore_index
        self.weight = weight

    def lr(self):
        return polynomial_lr(self.cfg["lr"], self.epoch, self.cfg["total_iters"], self.cfg["power"])

    def step(self, x_u8, target_u8):
        x = x_u8.to(torch.float32)  # S/train.py:121
        logits = unet_forward(self.p, x, self.variant, train=True)
        loss = cross_entropy(logits, target_u8.long(), self.weight, self.ignore_index)
        grads = torch.autograd.grad(loss, [self.p[k] for k in self.trainable])
        self.nstep += 1
        lr = self.lr()
        with torch.no_grad():
            for k, g in zip(self.trainable, grads):
                p = self.p[k]
                if self.cfg["kind"] == "sgd":
                    newp, buf = sgd_step(p, g, self.state.get(k), lr, self.cfg["momentum"],
                                         self.cfg["weight_decay"], k not in self.state)
                    self.state[k] = buf
                else:
                    m, v = self.state.get(k, (torch.zeros_like(p), torch.zeros_like(p)))
                    newp, m, v = adam_step(p, g, m, v, self.nstep, lr, self.cfg["betas"],
                                           self.cfg["eps"], self.cfg["weight_decay"])
                    self.state[k] = (m, v)
                p.copy_(newp)
        return float(loss.detach()), logits.detach(), dict(zip(self.trainable, grads))

    def end_epoch(self):
        self.epoch += 1  # scheduler.step(), S/train.py:132-133


# --------------------------------------------------------------------------- evaluation
def predict_labels(logits):
    """argmax over the class axis, ties -> lowest index (torch.max(dim=0) S/utils.py:90,
    torch.argmax S/utils.py:158)."""
    return torch.argmax(logits, dim=1)


def confusion_matrix(pred, target, num_classes=16, ignore_index=None):
    """torchmetrics MulticlassConfusionMatrix(num_classes, ignore_index) as used at
    S/utils.py:76-78,93-94: rows = target, cols = prediction, pixels whose TARGET equals
    ignore_index are dropped.  Restated as bincount(target*C + pred)."""
    p = np.asarray(pred).reshape(-1).astype(np.int64)
    t = np.asarray(target).reshape(-1).astype(np.int64)
    if ignore_index is not None:
        keep = t != ignore_index
        p, t = p[keep], t[keep]
    return np.bincount(t * num_classes + p, minlength=num_classes * num_classes) \
        .reshape(num_classes, num_classes).astype(np.int64)


def _class_scores(conf, formula, ignore):
    """S/utils.py:319-346."""
    scores, excluded = [], []
    C = conf.shape[1]
    for i in range(C):
        tp = float(conf[i, i])
        fp = float(conf[:, i].sum() - conf[i, i])
        fn = float(conf[i, :].sum() - conf[i, i])
        if ignore(tp, fp, fn):
            scores.append(0.0)
            excluded.append(i)
        else:
            scores.append(formula(tp, fp, fn))
    return scores, excluded


def _mean_excluding(scores, excluded):
    """S/utils.py:343-346: float32 tensor mean over the classes not excluded."""
    inc = np.array([s for i, s in enumerate(scores) if i not in excluded], dtype=np.float32)
    return float(inc.mean(dtype=np.float32)) if len(inc) else float("nan")


def metrics(conf):
    """print_metrics (S/utils.py:375-403): mIoU, mPrec, mRec, Dice, OA and per-class IoU."""
    conf = np.asarray(conf)
    iou, ex_iou = _class_scores(conf, lambda tp, fp, fn: tp / (tp + fn + fp), lambda tp, fp, fn: tp + fn == 0)
    prec, ex_p = _class_scores(conf, lambda tp, fp, fn: tp / (tp + fp), lambda tp, fp, fn: tp + fp == 0)
    rec, ex_r = _class_scores(conf, lambda tp, fp, fn: tp / (tp + fn), lambda tp, fp, fn: tp + fn == 0)
    f1, ex_f = _class_scores(conf, lambda tp, fp, fn: 2 * tp / (2 * tp + fn + fp), lambda tp, fp, fn: tp + fn == 0)
    return {
        "perclass_IoU": [float(np.float32(s)) for s in iou],
        "excluded": ex_iou,
        "mIoU": _mean_excluding(iou, ex_iou),
        "precision_score": _mean_excluding(prec, ex_p),
        "recall_score": _mean_excluding(rec, ex_r),
        "dice_score": _mean_excluding(f1, ex_f),
        "oa_score": float(np.trace(conf)) / float(conf.sum()),
    }


# --------------------------------------------------------------------------- numpy twins
def numpy_conv3x3(x, w, b, relu=False):
    """float64 direct 3x3 / padding 1 convolution (independent of ATen)."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    B, C, H, W = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    out = np.zeros((B, w.shape[0], H, W))
    for kh in range(3):
        for kw in range(3):
            out += np.einsum("bchw,oc->bohw", xp[:, :, kh:kh + H, kw:kw + W], w[:, :, kh, kw])
    out += np.asarray(b, np.float64)[None, :, None, None]
    return np.maximum(out, 0) if relu else out


def numpy_batch_norm_train(x, gamma, beta):
    x = np.asarray(x, np.float64)
    mean = x.mean(axis=(0, 2, 3), keepdims=True)
    var = x.var(axis=(0, 2, 3), keepdims=True)
    return (x - mean) / np.sqrt(var + BN_EPS) * np.asarray(gamma)[None, :, None, None] \
        + np.asarray(beta)[None, :, None, None]


# --------------------------------------------------------------------------- synthetic tiles
def synthetic_tiles(B, S, NC, seed=1234, structured=False):
    """u8 images in 0..255 (raw, un-normalised) and u8 labels in [0,NC).
    structured=True gives Voronoi label maps with per-class colours + noise (mIoU runs)."""
    g = torch.Generator().manual_seed(seed)
    if not structured:
        img = torch.randint(0, 256, (B, 3, S, S), dtype=torch.uint8, generator=g)
        lab = torch.randint(0, NC, (B, S, S), dtype=torch.uint8, generator=g)
        return img, lab
    base = torch.randint(30, 226, (NC, 3), generator=torch.Generator().manual_seed(99)).float()
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    imgs, labs = [], []
    for _ in range(B):
        n = int(torch.randint(8, 33, (1,), generator=g))
        sy = torch.randint(0, S, (n,), generator=g)
        sx = torch.randint(0, S, (n,), generator=g)
        cls = torch.randint(1, NC, (n,), generator=g)
        bg = torch.rand(n, generator=g) < 0.05  # ~5 % unlabeled cells
        cls = torch.where(bg, torch.zeros_like(cls), cls)
        d = (yy[None] - sy[:, None, None]) ** 2 + (xx[None] - sx[:, None, None]) ** 2
        lab = cls[d.argmin(0)]
        img = base[lab].permute(2, 0, 1) + 20.0 * torch.randn(3, S, S, generator=g)
        imgs.append(img.clamp(0, 255).to(torch.uint8))
        labs.append(lab.to(torch.uint8))
    return torch.stack(imgs), torch.stack(labs)
