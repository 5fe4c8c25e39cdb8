"""Model factory side of the drop-in: `Urnet` / `Urnetv2` with the reference's nn.Module contract
(source/scripts/nets.py:12-33: `requires_context`, `wrapper`, `returns_logits`, `forward(x, context=None)`),
the reference's `state_dict()` keys and shapes (nets.py:34-199, blocks.py:8-49), and a forward/backward that runs
entirely in the HIP library through `UNetEngine`.

The modules hold no torch compute: sub-modules exist only so that parameter names are the reference's
(`encode1.0.layer.0.weight`, `upscale1.0.weight`, `decode_forward4.1.bias`, ...) and reference checkpoints load.
All parameters are views of ONE flat f32 buffer (and all gradients of one flat gradient buffer): the fused
optimiser updates them in a single launch and data-parallel training all-reduces the same buffer in buckets.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from .engine import WIDTHS, UNetEngine
from .deeplab_engine import LOW_LEVEL_CHANNELS, DeepLabEngine
from .resnet_engine import ARCHS as RESNET_ARCHS, DECODER_CHANNELS, ResNetUNetEngine
from .swin_engine import HIDDEN as UPER_HIDDEN, POOL_SCALES, VARIANTS as SWIN_VARIANTS, SwinUPerNetEngine

PRECISIONS = {"fp32": torch.float32, "bf16": torch.bfloat16,
              # "fp8" = BASELINE configs[4] "mixed bf16/fp8 convs": bf16 storage, the qualifying 3x3 convolutions on the fp8 MFMA (ResNet-UNets)
              "fp8": torch.bfloat16}


def unet_param_spec(variant: str, num_classes: int):
    """(name, shape, kind) in the reference's registration order; kind in {conv_w, conv_b, bn_w, bn_b, rm, rv}."""
    spec = []

    def conv(p, cin, cout, k=3):
        spec.append((p + ".weight", (cout, cin, k, k), "conv_w"))
        spec.append((p + ".bias", (cout,), "conv_b"))

    def bn(p, c):
        spec.extend([(p + ".weight", (c,), "bn_w"), (p + ".bias", (c,), "bn_b"),
                     (p + ".running_mean", (c,), "rm"), (p + ".running_var", (c,), "rv"),
                     (p + ".num_batches_tracked", (), "nbt")])

    cin = 3
    for lvl, w in enumerate(WIDTHS, start=1):
        base = 0 if lvl == 1 else 1  # encode2..5 keep the reference's MaxPool2d at index 0 (nets.py:129-147)
        for j in range(2):
            p = f"encode{lvl}.{base + j}.layer"
            conv(p + ".0", cin if j == 0 else w, w)
            bn(p + ".1", w)
        cin = w
    for d in range(1, 5):
        cin = WIDTHS[5 - d]
        cout = cin // 2
        if variant == "Unetv2":
            spec.append((f"upscale{d}.0.weight", (cin, cout, 2, 2), "conv_w"))  # ConvTranspose2d: [Cin, Cout, 2, 2]
            spec.append((f"upscale{d}.0.bias", (cout,), "conv_b"))
        else:
            conv(f"upscale{d}.0.layer.1", cin, cout)
        p = f"decode_forward{d}.0.layer"
        conv(p + ".0", cin, cout)
        bn(p + ".2", cout)
        conv(p + ".3", cout, cout)
        bn(p + ".5", cout)
    conv("decode_forward4.1", 64, num_classes, k=1)
    return spec


class _Node(nn.Module):
    """bare container used to reproduce the reference's dotted parameter names."""


def _descend(root: nn.Module, dotted: str):
    parts = dotted.split(".")
    m = root
    for part in parts[:-1]:
        if part not in m._modules:
            m.add_module(part, _Node())
        m = m._modules[part]
    return m, parts[-1]


LAZY_HEAD = os.environ.get("CVCS_LAZY_HEAD", "1") == "1"
# attribute reads / methods that look at a tensor's metadata only: they do not make a LazyLogits handle run the head
_META_GETTERS = {"shape", "dtype", "device", "requires_grad", "grad_fn", "is_cuda", "ndim", "layout", "is_leaf", "names", "grad", "output_nr",
                 "_version", "is_sparse", "is_quantized", "is_meta", "is_mkldnn", "is_nested", "_backward_hooks", "retains_grad", "_grad_fn", "_grad"}
_META_METHODS = {"size", "dim", "stride", "numel", "nelement", "ndimension", "element_size", "is_contiguous", "storage_offset", "requires_grad_",
                 "register_hook", "retain_grad", "is_floating_point", "is_complex", "get_device", "_is_view", "is_shared", "has_names", "__len__",
                 "type", "is_same_size", "__format__"}


class LazyLogits(torch.Tensor):
    """The logits `net(x)` returns in train mode (S/train.py:121) as a HANDLE: the engine-owned [B, NC, S, S] f32 buffer whose content - the
    1x1 head's output - is computed the first time anything reads it.  `utils.CrossEntropyLoss` recognises an unread handle and launches
    cvcs_head_ce instead: head, softmax cross-entropy and their backward in one pass over the last decoder activation, so that in the
    reference's loop `logits = net(x); loss = criterion(logits, y); loss.backward()` the logits and their gradient never exist in memory.
    Every other use - printing, `.argmax(1)`, a user's own loss, `.cpu()` - sees an ordinary tensor."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        meta = (name == "__get__" and getattr(getattr(func, "__self__", None), "__name__", "") in _META_GETTERS) or name in _META_METHODS
        if not meta:
            for a in args:
                if isinstance(a, LazyLogits):
                    a.materialize()
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)

    def materialize(self):
        st = self.__dict__.get("_cvcs_lazy")
        if st is not None and not st["mat"]:
            st["mat"] = True
            if st["serial"] == st["engine"].fwd_serial:      # (a later forward has overwritten the activations: the handle is stale, as the buffer was before)
                st["engine"].materialize_logits()


def lazy_state(t):
    """the state of a LazyLogits handle that nobody has read or consumed yet and whose forward pass is still the engine's last one, else None.
    A handle whose loss was fused still materialises when somebody reads it afterwards (the activations live until the next forward)."""
    if type(t) is not LazyLogits:
        return None
    st = t.__dict__.get("_cvcs_lazy")
    if st is None or st["mat"] or st["fused"] or st["serial"] != st["engine"].fwd_serial:
        return None
    return st


def lazy_fused(t) -> bool:
    return type(t) is LazyLogits and bool(t.__dict__.get("_cvcs_lazy", {}).get("fused"))


class _UNetFunction(torch.autograd.Function):
    """glue so that the reference's `loss.backward()` (S/train.py:125) reaches the HIP backward pass."""

    @staticmethod
    def forward(ctx, anchor, x, net):
        ctx.net = net
        eng = net._engine
        shape = (x.shape[0], net.num_classes, x.shape[2], x.shape[3])
        if LAZY_HEAD and getattr(eng, "lazy_head_ok", None) is not None and eng.lazy_head_ok():
            buf = eng.forward(x, train=True, lazy=True).view(-1).view(shape)
            t = torch.Tensor._make_subclass(LazyLogits, buf, False)
            t.__dict__["_cvcs_lazy"] = dict(engine=eng, serial=eng.fwd_serial, mat=False, fused=False)
            return t
        # the returned tensor aliases an engine-owned buffer: valid until the next forward of this network
        return eng.forward(x, train=True).view(-1).view(shape)

    @staticmethod
    def backward(ctx, dlogits):
        net = ctx.net
        net._engine.backward(dlogits)
        net._publish_grads()
        return None, None, None


class _HipUNet(nn.Module):
    variant = "Unetv2"

    def __init__(self, num_classes: int, precision: str = "bf16"):
        super().__init__()
        assert precision in PRECISIONS, f"precision must be one of {list(PRECISIONS)}"
        assert precision != "fp8" or self.variant in ("ResnetUnet", "SwinTUperNet", "SwinBUperNet", "Unet", "Unetv2"), \
            "precision 'fp8' (mixed bf16 / fp8 convolutions) is built for the U-Nets, the ResNet-UNets and the UPerNet head of Swin + UPerNet"
        self.requires_context = False   # S/nets.py:37,120
        self.wrapper = False
        self.returns_logits = True
        self.num_classes = num_classes
        self.precision = precision
        self._spec = self._build_spec()
        self._engine = None
        self._flat = None
        self._flat_grad = None
        self._nbt_pending = 0
        g = torch.Generator().manual_seed(torch.initial_seed() % (2 ** 31))
        for name, shape, kind in self._spec:
            holder, leaf = _descend(self, name)
            if kind == "conv_w":
                fan_in = shape[1] * shape[2] * shape[3]
                bound = 1.0 / math.sqrt(fan_in)          # torch's default kaiming_uniform(a=sqrt(5)) bound
                holder.register_parameter(leaf, nn.Parameter((torch.rand(shape, generator=g) * 2 - 1) * bound))
            elif kind == "conv_b":
                w = holder._parameters["weight"]
                bound = 1.0 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3])
                holder.register_parameter(leaf, nn.Parameter((torch.rand(shape, generator=g) * 2 - 1) * bound))
            elif kind == "lin_w":      # nn.Linear / relative-position table: truncated-normal(0.02)-like small uniform
                holder.register_parameter(leaf, nn.Parameter((torch.rand(shape, generator=g) * 2 - 1) * 0.035))
            elif kind == "lin_b":
                holder.register_parameter(leaf, nn.Parameter(torch.zeros(shape)))
            elif kind == "bn_w":
                holder.register_parameter(leaf, nn.Parameter(torch.ones(shape)))
            elif kind == "bn_b":
                holder.register_parameter(leaf, nn.Parameter(torch.zeros(shape)))
            elif kind == "rm":
                holder.register_buffer(leaf, torch.zeros(shape))
            elif kind == "rv":
                holder.register_buffer(leaf, torch.ones(shape))
            else:
                holder.register_buffer(leaf, torch.tensor(0, dtype=torch.long))

    def _build_spec(self):
        return unet_param_spec(self.variant, self.num_classes)

    def _build_engine(self, dev):
        return UNetEngine(self.variant, self.num_classes, PRECISIONS[self.precision], dev, fp8=self.precision == "fp8")

    # ------------------------------------------------------------------------------------------------ flat storage
    def _flat_order(self, params):
        """order of the parameters inside the flat buffers (an internal layout: state_dict / named_parameters are unaffected)"""
        return params

    def _ensure_flat(self):
        """(re)build the flat f32 parameter / gradient buffers on the parameters' current device and re-point every
        nn.Parameter at its slice (needed after `.to(device)` / `load_state_dict`, both of which keep the objects)."""
        params = self._flat_order(OrderedDict(self.named_parameters()))
        first = next(iter(params.values()))
        dev = first.device
        if dev.type != "cuda":
            raise RuntimeError("cvcs_amd networks run on the GPU only (config device: gpu); there is no CPU path. "
                               "Move the network with .to('cuda:0') as the reference's load_network does.")
        total = sum(_align4(p.numel()) for p in params.values())
        ok = self._flat is not None and self._flat.device == dev and self._flat.numel() == total
        if ok:
            off = 0
            for p in params.values():
                if p.data_ptr() != self._flat.data_ptr() + off * 4:
                    ok = False
                    break
                off += _align4(p.numel())
        if not ok:
            flat = torch.zeros(total, dtype=torch.float32, device=dev)
            off = 0
            for p in params.values():
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1).float())
                p.data = flat[off:off + n].view(p.shape)
                off += _align4(n)
            self._flat = flat
            self._flat_grad = torch.zeros_like(flat)
            self._engine = None
        if self._engine is None:
            eng = self._build_engine(dev)
            P, G, off = OrderedDict(), OrderedDict(), 0
            for name, p in params.items():
                n = p.numel()
                P[name] = p.data
                G[name] = self._flat_grad[off:off + n].view(p.shape)
                off += _align4(n)
            bufs = {k: v for k, v in self.named_buffers() if v.dtype == torch.float32}
            eng.bind(P, G, bufs)
            self._engine = eng
            self._grad_views = G

    def _publish_grads(self):
        for name, p in self.named_parameters():
            p.grad = self._grad_views[name]

    def flat_parameters(self):
        self._ensure_flat()
        return self._flat, self._flat_grad

    # ------------------------------------------------------------------------------------------------ nn.Module API
    def forward(self, x: torch.Tensor, context: torch.Tensor = None):
        """x: [B,3,S,S] u8 or f32 (raw 0..255, S/train.py:121) on the GPU -> f32 logits [B,NC,S,S]."""
        self._ensure_flat()
        if x.dtype not in (torch.uint8, torch.float32):
            x = x.float()
        if self.training:
            self._nbt_pending += 1   # BatchNorm2d.num_batches_tracked, materialised lazily in state_dict()
            if torch.is_grad_enabled():
                anchor = next(self.parameters())
                return _UNetFunction.apply(anchor, x, self)
            return self._engine.forward(x, train=True).clone()
        return self._engine.forward(x, train=False).clone()

    @torch.no_grad()
    def predict_labels(self, x: torch.Tensor) -> torch.Tensor:
        """u8 labels [B,S,S] = argmax over classes of the eval-mode logits (ties -> lowest class, as torch.max /
        torch.argmax at S/utils.py:90,158): the head and the argmax are one launch, the logits are never written.
        Running statistics are used whatever the module's train / eval flag says."""
        self._ensure_flat()
        if x.dtype not in (torch.uint8, torch.float32):
            x = x.float()
        labels = torch.empty((x.shape[0], x.shape[2], x.shape[3]), dtype=torch.uint8, device=x.device)
        return self._engine.forward_labels(x, labels)

    def state_dict(self, *args, **kwargs):
        if self._nbt_pending:
            for name, b in self.named_buffers():
                if name.endswith("num_batches_tracked"):
                    b += self._nbt_pending
            self._nbt_pending = 0
        return super().state_dict(*args, **kwargs)


def _align4(n: int) -> int:
    return (n + 3) // 4 * 4


class Urnet(_HipUNet):
    """U-Net with bilinear x2 + conv3x3 up-sampling (S/nets.py:34-115)."""
    variant = "Unet"


class Urnetv2(_HipUNet):
    """U-Net with ConvTranspose2d(k2,s2) up-sampling (S/nets.py:117-199)."""
    variant = "Unetv2"


# ---------------------------------------------------------------------------------------------------- ResNet-encoder U-Nets
GN_DECODER_CHANNELS = (256, 128, 64, 64, 64)      # GroupNorm(32) needs multiples of 32 channels: the gn_silu decoder keeps the wide last stages


def default_decoder_channels(decoder_norm="bn_relu"):
    return DECODER_CHANNELS if decoder_norm == "bn_relu" else GN_DECODER_CHANNELS


def resnet_unet_param_spec(arch: str, num_classes: int, decoder_channels=None, decoder_norm="bn_relu"):
    """(name, shape, kind) in forward order.  Encoder names are torchvision's ResNet names under `encoder.` (ImageNet /
    torchvision checkpoints map one to one); decoder / head names follow the common `decoder.blocks.N.convK.{0,1}` /
    `segmentation_head.0` scheme.  No conv of the encoder or decoder has a bias (each is followed by a BatchNorm)."""
    kind, depths, widths = RESNET_ARCHS[arch]
    decoder_channels = tuple(decoder_channels) if decoder_channels is not None else default_decoder_channels(decoder_norm)
    spec = []

    def conv(p, cin, cout, k, bias=False):
        spec.append((p + ".weight", (cout, cin, k, k), "conv_w"))
        if bias:
            spec.append((p + ".bias", (cout,), "conv_b"))

    def bn(p, c):
        spec.extend([(p + ".weight", (c,), "bn_w"), (p + ".bias", (c,), "bn_b"), (p + ".running_mean", (c,), "rm"),
                     (p + ".running_var", (c,), "rv"), (p + ".num_batches_tracked", (), "nbt")])

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
    def dnorm(p, c):   # decoder_norm="gn_silu": GroupNorm(32) affine only - no running statistics
        if decoder_norm == "gn_silu":
            spec.extend([(p + ".weight", (c,), "bn_w"), (p + ".bias", (c,), "bn_b")])
        else:
            bn(p, c)

    for i, (dc, sk) in enumerate(zip(decoder_channels, skips)):
        p = f"decoder.blocks.{i}"
        conv(p + ".conv1.0", cin + sk, dc, 3); dnorm(p + ".conv1.1", dc)
        conv(p + ".conv2.0", dc, dc, 3); dnorm(p + ".conv2.1", dc)
        cin = dc
    conv("segmentation_head.0", cin, num_classes, 1, bias=True)
    return spec


class ResnetUnet(_HipUNet):
    """U-Net with a ResNet encoder (BASELINE.json configs 1, 2, 5) behind the reference's nn.Module contract
    (S/nets.py:12-33) and factory seam (S/utils.py:174-195): ResNet v1.5 encoder, bilinear-upsample decoder of
    (conv3x3 -> BN -> ReLU) pairs over [up-sampled | encoder feature], 1x1 head.  Input: the raw 0..255 tile, as every
    network of the reference gets it (S/train.py:121).  The tile side must be a multiple of 32."""
    arch = "resnet50"
    variant = "ResnetUnet"

    def __init__(self, num_classes: int, precision: str = "bf16", decoder_norm: str = "bn_relu", decoder_channels=None):
        """decoder_norm: "bn_relu" (default: conv -> BatchNorm -> ReLU) | "gn_silu" (conv -> GroupNorm(32) -> SiLU, normalisation and
        activation fused in one pass; config key `decoder_norm`).  decoder_channels: the five decoder widths (config key `decoder_channels`;
        default (256, 128, 64, 32, 16), the common public U-Net-on-ResNet layout; multiples of 16, the last one 16 / 32 / 64)"""
        assert decoder_norm in ("bn_relu", "gn_silu")
        self.decoder_norm = decoder_norm
        self.decoder_channels = tuple(decoder_channels) if decoder_channels is not None else default_decoder_channels(decoder_norm)
        super().__init__(num_classes, precision)

    def _build_spec(self):
        return resnet_unet_param_spec(self.arch, self.num_classes, self.decoder_channels, decoder_norm=self.decoder_norm)

    def _build_engine(self, dev):
        return ResNetUNetEngine(self.arch, self.num_classes, PRECISIONS[self.precision], dev, decoder_channels=self.decoder_channels,
                                decoder_norm=self.decoder_norm, fp8=self.precision == "fp8")


class Resnet18Unet(ResnetUnet):
    arch = "resnet18"


class Resnet34Unet(ResnetUnet):
    arch = "resnet34"


class Resnet50Unet(ResnetUnet):
    arch = "resnet50"


# ---------------------------------------------------------------------------------------------------- DeepLab family
def deeplab_param_spec(arch: str, num_classes: int, plus: bool = True):
    """(name, shape, kind): the ResNet encoder (torchvision names under `encoder.`), ASPP (`aspp.convs.N.*`, `aspp.project.*`
    as in torchvision.models.segmentation.deeplabv3), then the V3+ decoder (`low_level`, `decoder.conv1/2`) or the V3 head
    (`head`), and the 1x1 `classifier` (the reference's replaced `classifier[4]`, S/nets.py:243-244)."""
    spec = [e for e in resnet_unet_param_spec(arch, num_classes) if e[0].startswith("encoder.")]
    widths = RESNET_ARCHS[arch][2]

    def conv(p, cin, cout, k, bias=False):
        spec.append((p + ".weight", (cout, cin, k, k), "conv_w"))
        if bias:
            spec.append((p + ".bias", (cout,), "conv_b"))

    def bn(p, c):
        spec.extend([(p + ".weight", (c,), "bn_w"), (p + ".bias", (c,), "bn_b"), (p + ".running_mean", (c,), "rm"),
                     (p + ".running_var", (c,), "rv"), (p + ".num_batches_tracked", (), "nbt")])

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


class DeepLabV3Plus(_HipUNet):
    """DeepLabV3+ with a ResNet-50 encoder at output stride 16, ASPP rates 6/12/18 (BASELINE.json configs[2]) behind the
    reference's nn.Module contract; see deeplab_engine.py.  The tile side must be a multiple of 32; train mode needs a batch
    of at least 2 tiles (the image-pooling branch normalises a 1x1 map over the batch, as in torchvision)."""
    arch, output_stride, plus = "resnet50", 16, True
    variant = "DeepLab"
    default_aspp_dropout = 0.0     # (BASELINE names the architecture only; config key `aspp_dropout` switches it on)

    def __init__(self, num_classes: int, precision: str = "bf16", aspp_dropout=None, dropout_seed: int = 0):
        """aspp_dropout: p of the nn.Dropout behind the ASPP projection in train mode (None: the class default - 0.5 for the reference's
        torchvision DeepLabV3 wrappers, as torchvision's DeepLabHead has it; 0 for DeepLabV3+).  The mask is counter-based:
        (dropout_seed, step, element) -> keep, replayable on the host (ops.dropout_mask_host)."""
        self.aspp_dropout = self.default_aspp_dropout if aspp_dropout is None else float(aspp_dropout)
        self.dropout_seed = int(dropout_seed)
        super().__init__(num_classes, precision)

    def _build_spec(self):
        return deeplab_param_spec(self.arch, self.num_classes, self.plus)

    def _build_engine(self, dev):
        return DeepLabEngine(self.arch, self.num_classes, PRECISIONS[self.precision], dev, self.output_stride, self.plus,
                             aspp_dropout=self.aspp_dropout, dropout_seed=self.dropout_seed)


class _TorchvisionDeepLabKeys:
    """`state_dict()` / `load_state_dict()` in the key names of the reference's wrapper modules (S/nets.py:234-275: `self.model =
    torchvision deeplabv3_resnet*`, so a reference checkpoint reads `model.backbone.layer3.0.conv1.weight`,
    `model.classifier.0.convs.1.0.weight`, `model.classifier.4.bias` ...).
    Reference -> here: loads as is (`module.` prefix accepted).  torchvision's auxiliary FCN head (`model.aux_classifier.*`, present when
    the reference starts from the COCO weights; its forward never uses it, S/nets.py:248-250) takes no part in this network: its
    tensors are KEPT as loaded and written back by `state_dict()`, so a reference checkpoint survives a round trip through this class
    key for key and loads back into the reference's module strictly.
    Here -> reference, for a network that never saw a reference checkpoint: the aux head's keys are absent, so the reference's
    `custom_load` (a strict `load_state_dict`, S/nets.py:252-257) needs `strict=False` for them (or `aux_loss=None` at construction);
    the torch-format optimiser state covers this network's parameters only (the reference's optimiser also holds the aux head's), so the
    reference starts its optimiser fresh.  Train mode applies the nn.Dropout(0.5) torchvision's ASPP ends with (see `aspp_dropout`)."""
    default_aspp_dropout = 0.5
    _aux_state = None
    _KEYS = (("encoder.", "model.backbone."), ("aspp.convs.", "model.classifier.0.convs."), ("aspp.project.", "model.classifier.0.project."),
             ("head.0.", "model.classifier.1."), ("head.1.", "model.classifier.2."), ("classifier.", "model.classifier.4."))

    @classmethod
    def _to_reference(cls, k):
        for mine, ref in cls._KEYS:
            if k.startswith(mine):
                return ref + k[len(mine):]
        return k

    @classmethod
    def _from_reference(cls, k):
        if k.startswith("module."):          # DataParallel prefix (S/nets.py:252-257)
            k = k[len("module."):]
        for mine, ref in cls._KEYS:
            if k.startswith(ref):
                return mine + k[len(ref):]
        return k

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        out = type(sd)((self._to_reference(k), v) for k, v in sd.items())
        if self._aux_state:
            out.update(self._aux_state)
        if hasattr(sd, "_metadata"):
            out._metadata = sd._metadata
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kwargs):
        aux = OrderedDict((k[len("module."):] if k.startswith("module.") else k, v.detach().clone()) for k, v in state_dict.items()
                          if ".aux_classifier." in k)
        self._aux_state = aux or None
        sd = OrderedDict((self._from_reference(k), v) for k, v in state_dict.items() if ".aux_classifier." not in k)
        return super().load_state_dict(sd, strict=strict, **kwargs)

    def custom_load(self, checkpoint):       # S/nets.py:252-257, called by the reference's load_checkpoint when net.wrapper
        self.load_state_dict(checkpoint["model_state_dict"])


class DeepLabv3Resnet101(_TorchvisionDeepLabKeys, DeepLabV3Plus):
    """the network behind the reference's factory name `Resnet101` (S/nets.py:234-257: torchvision deeplabv3_resnet101 with
    `classifier[4]` -> Conv2d(256, num_classes, 1), output stride 8, ASPP rates 12/24/36, forward returns d['out']), built on
    the HIP kernels, with the reference module's `wrapper` / `custom_load` contract and state_dict key names.  The reference loads
    COCO weights through a network fetch; here the weights are randomly initialised unless a checkpoint is loaded."""
    arch, output_stride, plus = "resnet101", 8, False

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.wrapper = True


class DeepLabv3Resnet50(_TorchvisionDeepLabKeys, DeepLabV3Plus):
    """S/nets.py:259-275 (same head on a ResNet-50; not reachable from the reference's load_network either)"""
    arch, output_stride, plus = "resnet50", 8, False

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.wrapper = True


def mobilenet_param_spec(num_classes: int):
    """(name, shape, kind) of torchvision's deeplabv3_mobilenet_v3_large as the reference wraps it (S/nets.py:277-311): the
    `mobilenet_v3_large(dilated=True).features` backbone under `encoder.N...` (torchvision: `backbone.N...`; block layers as
    `encoder.N.block.M.{0,1}` / `.fc1` / `.fc2`), then the DeepLabHead names of deeplab_param_spec (`aspp.*`, `head.*`, `classifier`)."""
    from .mobilenet_engine import BLOCKS, LAST_CHANNELS, block_layers, make_divisible
    spec = []

    def conv(p, cin, cout, k, bias=False, groups=1):
        spec.append((p + ".weight", (cout, cin // groups, k, k), "conv_w"))
        if bias:
            spec.append((p + ".bias", (cout,), "conv_b"))

    def bn(p, c):
        spec.extend([(p + ".weight", (c,), "bn_w"), (p + ".bias", (c,), "bn_b"), (p + ".running_mean", (c,), "rm"),
                     (p + ".running_var", (c,), "rv"), (p + ".num_batches_tracked", (), "nbt")])

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
    last = f"encoder.{len(BLOCKS) + 1}"
    conv(last + ".0", BLOCKS[-1][3], LAST_CHANNELS, 1); bn(last + ".1", LAST_CHANNELS)
    cin = LAST_CHANNELS
    conv("aspp.convs.0.0", cin, 256, 1); bn("aspp.convs.0.1", 256)
    for i in (1, 2, 3):
        conv(f"aspp.convs.{i}.0", cin, 256, 3); bn(f"aspp.convs.{i}.1", 256)
    conv("aspp.convs.4.1", cin, 256, 1); bn("aspp.convs.4.2", 256)
    conv("aspp.project.0", 5 * 256, 256, 1); bn("aspp.project.1", 256)
    conv("head.0", 256, 256, 3); bn("head.1", 256)
    conv("classifier", 256, num_classes, 1, bias=True)
    return spec


class DeepLabV3MobileNet(_TorchvisionDeepLabKeys, DeepLabV3Plus):
    """the network behind the reference's factory name `MobileNet` (S/nets.py:277-311, S/utils.py:183-184: torchvision
    deeplabv3_mobilenet_v3_large with `classifier[4]` -> Conv2d(256, num_classes, 1); forward casts to float and returns d['out']) on the
    HIP kernels (mobilenet_engine.py), with the reference module's `wrapper` / `custom_load` contract (:306-311) and state_dict key names
    (`model.backbone.N...`, `model.classifier...`; `model.aux_classifier.*` tensors of a reference checkpoint are kept as loaded).  The
    reference starts from COCO weights through a network fetch; here the weights are randomly initialised unless a checkpoint is loaded.
    Its `_googlenet_backbone` / `_resnet18_backbone` switches (never set by the factory) are not built.  The tile side must be a multiple of
    32; train mode needs at least 2 tiles (the image-pooling BatchNorm).  The input is fed raw, as the reference does."""
    arch, output_stride, plus = "mobilenet_v3_large", 16, False
    variant = "MobileNet"

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.wrapper = True

    def _build_spec(self):
        return mobilenet_param_spec(self.num_classes)

    def _build_engine(self, dev):
        from .mobilenet_engine import MobileNetEngine
        return MobileNetEngine(self.num_classes, PRECISIONS[self.precision], dev, aspp_dropout=self.aspp_dropout, dropout_seed=self.dropout_seed)


# ---------------------------------------------------------------------------------------------------- Swin-T + UPerNet
def swin_upernet_param_spec(num_classes: int, variant: str = "tiny"):
    """(name, shape, kind): the parameter names of transformers' UperNetForSemanticSegmentation over a SwinBackbone (embed 96, depths
    2/2/6/2, heads 3/6/12/24, window 7; UPerNet hidden 512, pool scales 1/2/3/6), so its checkpoints load by name.  Within an
    attention block the q / k / v weights, then their biases, are registered ADJACENTLY: in the flat parameter buffer they form one
    [3C, C] matrix and one [3C] bias, which the engine runs as a single GEMM."""
    embed, SWIN_DEPTHS, SWIN_HEADS = SWIN_VARIANTS[variant]      # noqa: N806
    SWIN_DIMS = tuple(embed * 2 ** i for i in range(4))          # noqa: N806
    spec = []

    def lin(p, cin, cout, bias=True):
        spec.append((p + ".weight", (cout, cin), "lin_w"))
        if bias:
            spec.append((p + ".bias", (cout,), "lin_b"))

    def ln(p, c):
        spec.extend([(p + ".weight", (c,), "bn_w"), (p + ".bias", (c,), "bn_b")])

    def cbr(p, cin, cout, k):
        spec.append((p + ".conv.weight", (cout, cin, k, k), "conv_w"))
        spec.extend([(p + ".batch_norm.weight", (cout,), "bn_w"), (p + ".batch_norm.bias", (cout,), "bn_b"),
                     (p + ".batch_norm.running_mean", (cout,), "rm"), (p + ".batch_norm.running_var", (cout,), "rv"),
                     (p + ".batch_norm.num_batches_tracked", (), "nbt")])

    pe = "backbone.swin.embeddings.patch_embeddings.projection"
    spec.append((pe + ".weight", (SWIN_DIMS[0], 3, 4, 4), "conv_w"))
    spec.append((pe + ".bias", (SWIN_DIMS[0],), "conv_b"))
    ln("backbone.swin.embeddings.norm", SWIN_DIMS[0])
    for s, (depth, dim, heads) in enumerate(zip(SWIN_DEPTHS, SWIN_DIMS, SWIN_HEADS)):
        for b in range(depth):
            p = f"backbone.swin.encoder.layers.{s}.blocks.{b}"
            for n in ("q_proj", "k_proj", "v_proj"):
                spec.append((f"{p}.attention.{n}.weight", (dim, dim), "lin_w"))
            for n in ("q_proj", "k_proj", "v_proj"):
                spec.append((f"{p}.attention.{n}.bias", (dim,), "lin_b"))
            lin(f"{p}.attention.o_proj", dim, dim)
            spec.append((f"{p}.attention.relative_position_bias.relative_position_bias_table", (169, heads), "lin_w"))
            ln(p + ".layernorm_before", dim)
            ln(p + ".layernorm_after", dim)
            lin(p + ".mlp.fc1", dim, 4 * dim)
            lin(p + ".mlp.fc2", 4 * dim, dim)
        if s < 3:
            lin(f"backbone.swin.encoder.layers.{s}.downsample.reduction", 4 * dim, 2 * dim, bias=False)
            ln(f"backbone.swin.encoder.layers.{s}.downsample.norm", 4 * dim)
    for s, dim in enumerate(SWIN_DIMS, start=1):
        ln(f"backbone.hidden_states_norms.stage{s}", dim)
    for i in range(len(POOL_SCALES)):
        cbr(f"decode_head.psp_modules.{i}.1", SWIN_DIMS[3], UPER_HIDDEN, 1)
    cbr("decode_head.bottleneck", SWIN_DIMS[3] + len(POOL_SCALES) * UPER_HIDDEN, UPER_HIDDEN, 3)
    for i in range(3):
        cbr(f"decode_head.lateral_convs.{i}", SWIN_DIMS[i], UPER_HIDDEN, 1)
    for i in range(3):
        cbr(f"decode_head.fpn_convs.{i}", UPER_HIDDEN, UPER_HIDDEN, 3)
    cbr("decode_head.fpn_bottleneck", 4 * UPER_HIDDEN, UPER_HIDDEN, 3)
    spec.append(("decode_head.classifier.weight", (num_classes, UPER_HIDDEN, 1, 1), "conv_w"))
    spec.append(("decode_head.classifier.bias", (num_classes,), "conv_b"))
    return spec


class SwinTUperNet(_HipUNet):
    """Swin-T encoder + UPerNet decoder (BASELINE.json configs[3]; the reference documents `TSwin` at configs/train/README.txt:44
    without a definition) behind the reference's nn.Module contract; see swin_engine.py.  Stochastic depth / dropout are off.  The
    tile side must be a multiple of 32; train mode needs a batch of at least 2 tiles (the 1x1 pyramid-pooling branch normalises a
    1x1 map over the batch).  Raw 0..255 inputs like every network of the reference (S/train.py:121)."""
    variant = "SwinTUperNet"
    swin = "tiny"

    def _build_spec(self):
        return swin_upernet_param_spec(self.num_classes, self.swin)

    def _build_engine(self, dev):
        return SwinUPerNetEngine(self.num_classes, PRECISIONS[self.precision], dev, self.swin, fp8=self.precision == "fp8")

    def _flat_order(self, params):
        # (1) q | k | v weights, then their biases, adjacent: the engine runs the three projections as one [3C, C] GEMM;
        # (2) the output LayerNorm of stage s right behind that stage's blocks (registered after the whole encoder, its gradient is
        #     finished between the downsample's and the blocks' - the data-parallel reducer takes "every gradient at a flat offset >= x
        #     is final" signals, so the flat order must be the reverse of the order the backward finishes gradients in)
        stage_norm = {s: [n for n in params if n.startswith(f"backbone.hidden_states_norms.stage{s + 1}.")] for s in range(3)}
        out = OrderedDict()
        for name, p in params.items():
            for s in range(3):
                if name == f"backbone.swin.encoder.layers.{s}.downsample.reduction.weight":
                    for n in stage_norm[s]:
                        out[n] = params[n]
            if any(name in v for v in stage_norm.values()):
                continue
            if name.endswith(".attention.q_proj.weight"):
                att = name[:-len("q_proj.weight")]
                for leaf in ("weight", "bias"):
                    for n in ("q_proj", "k_proj", "v_proj"):
                        out[f"{att}{n}.{leaf}"] = params[f"{att}{n}.{leaf}"]
            elif ".attention.q_proj." in name or ".attention.k_proj." in name or ".attention.v_proj." in name:
                continue
            else:
                out[name] = p
        assert len(out) == len(params)
        return out


class SwinBUperNet(SwinTUperNet):
    """Swin-B encoder (embed 128, depths 2/2/18/2, heads 4/8/16/32) + the same UPerNet decoder: the `BSwin` name of the reference's
    documentation (configs/train/README.txt:45)"""
    variant = "SwinBUperNet"
    swin = "base"


# ------------------------------------------------------------------------------------------------ SegformerMod
def segformer_param_spec(num_classes: int, variant: str = "b0"):
    """(name, shape, kind) under the reference wrapper's key names (S/nets.py:313-349: `self.segformer` = transformers'
    SegformerForSemanticSegmentation, `self.seq` = the ConvTranspose / ConvTranspose / Conv tail), transformers 5.x module layout.  Key and
    value projection weights, then their biases, are registered adjacently: the engine runs them as one [2C, C] GEMM."""
    from .segformer_engine import CONFIGS, ENC, HEAD
    cfg = CONFIGS[variant]
    spec = []

    def lin(p, cin, cout):
        spec.extend([(p + ".weight", (cout, cin), "lin_w"), (p + ".bias", (cout,), "lin_b")])

    def ln(p, c):
        spec.extend([(p + ".weight", (c,), "bn_w"), (p + ".bias", (c,), "bn_b")])

    def conv(p, cin, cout, k, kind="lin_w", bias=True):
        spec.append((p + ".weight", (cout, cin, k, k), kind))
        if bias:
            spec.append((p + ".bias", (cout,), "lin_b" if kind == "lin_w" else "conv_b"))

    cin = 3
    for s, (c, depth, sr, k) in enumerate(zip(cfg["hidden"], cfg["depths"], cfg["sr"], cfg["patch"])):
        conv(f"{ENC}.{s}.patch_embeddings.proj", cin, c, k)
        ln(f"{ENC}.{s}.patch_embeddings.layer_norm", c)
        for b in range(depth):
            p = f"{ENC}.{s}.blocks.{b}"
            ln(p + ".layernorm_before", c)
            lin(p + ".attention.q_proj", c, c)
            for leaf, shape, kind in (("weight", (c, c), "lin_w"), ("bias", (c,), "lin_b")):
                for n in ("k_proj", "v_proj"):
                    spec.append((f"{p}.attention.{n}.{leaf}", shape, kind))
            lin(p + ".attention.o_proj", c, c)
            if sr > 1:
                conv(p + ".attention.sequence_reduction.sequence_reduction", c, c, sr)
                ln(p + ".attention.sequence_reduction.layer_norm", c)
            ln(p + ".layernorm_after", c)
            ch = cfg["mlp_ratio"] * c
            lin(p + ".mlp.fc1", c, ch)
            spec.extend([(p + ".mlp.dwconv.dwconv.weight", (ch, 1, 3, 3), "lin_w"), (p + ".mlp.dwconv.dwconv.bias", (ch,), "lin_b")])
            lin(p + ".mlp.fc2", ch, c)
        ln(f"{ENC}.{s}.layer_norm", c)
        cin = c
    D = cfg["decoder"]
    for s, c in enumerate(cfg["hidden"]):
        lin(f"{HEAD}.linear_projections.{s}.proj", c, D)
    conv(HEAD + ".linear_fuse", D * len(cfg["hidden"]), D, 1, bias=False)
    spec.extend([(HEAD + ".batch_norm.weight", (D,), "bn_w"), (HEAD + ".batch_norm.bias", (D,), "bn_b"), (HEAD + ".batch_norm.running_mean", (D,), "rm"),
                 (HEAD + ".batch_norm.running_var", (D,), "rv"), (HEAD + ".batch_norm.num_batches_tracked", (), "nbt")])
    conv(HEAD + ".classifier", D, num_classes, 1, kind="conv_w")
    NC = num_classes
    # nn.ConvTranspose2d weights are [in, out, k, k]; torch's default init takes its fan-in from dim 1, as for Conv2d
    spec.extend([("seq.0.weight", (NC, NC, 8, 8), "conv_w"), ("seq.0.bias", (NC,), "conv_b"), ("seq.2.weight", (NC, NC, 4, 4), "conv_w"),
                 ("seq.2.bias", (NC,), "conv_b")])
    conv("seq.4", NC, NC, 3, kind="conv_w")
    return spec


_SEGFORMER_V4 = [
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


class SegformerMod(_HipUNet):
    """The reference's `SegformerMod` (S/nets.py:313-349; `load_network` name at S/utils.py:191-192): transformers' Segformer (Mix
    Transformer encoder, all-MLP decode head) with an NC-class classifier, followed by the reference's ConvTranspose2d(NC,NC,8,2,3) - ReLU -
    ConvTranspose2d(NC,NC,4,2,1) - ReLU - Conv2d(NC,NC,3,padding=1) tail; the input is normalised with the ImageNet constants on the RAW
    0..255 scale, as the reference's preprocessor does.  `pretrained=False` there means `SegformerConfig()` = MiT-b0, which is what this class
    builds (`variant="b3"`: the architecture of the checkpoint the reference fetches with pretrained=True - hidden 64 / 128 / 320 / 512, depths
    3 / 4 / 18 / 3, 64-wide heads, decoder 768; there is no network here to fetch it, a state_dict with its tensors loads by name).  Train mode applies the library's DropPath (0 ... 0.1 over the blocks) and the head's
    Dropout(0.1) from a counter-based generator (`stochastic=False`: off).  State-dict keys follow transformers 5.x; checkpoints written
    under the 4.x module layout (`segformer.segformer.encoder.block.N.M.attention.self.query...`) are mapped on load.  The tile side must be
    a multiple of 32, NC at most 32."""
    variant = "SegformerMod"
    mit = "b0"

    def __init__(self, num_classes: int, precision: str = "bf16", stochastic: bool = True, seed: int = 0, variant: str | None = None):
        if variant is not None:
            self.mit = variant
        self.stochastic, self.seed = bool(stochastic), int(seed)
        super().__init__(num_classes, precision)
        self.wrapper = True           # S/nets.py:318: load_checkpoint goes through custom_load

    def _build_spec(self):
        return segformer_param_spec(self.num_classes, self.mit)

    def _build_engine(self, dev):
        from .segformer_engine import SegformerEngine
        return SegformerEngine(self.num_classes, PRECISIONS[self.precision], dev, self.mit, stochastic=self.stochastic, seed=self.seed)

    def _flat_order(self, params):
        # key | value projection weights, then their biases, adjacent: the engine runs the two projections as one [2C, C] GEMM
        out = OrderedDict()
        for name, p in params.items():
            if name.endswith(".attention.k_proj.weight"):
                att = name[:-len("k_proj.weight")]
                for leaf in ("weight", "bias"):
                    for n in ("k_proj", "v_proj"):
                        out[f"{att}{n}.{leaf}"] = params[f"{att}{n}.{leaf}"]
            elif ".attention.k_proj." in name or ".attention.v_proj." in name:
                continue
            else:
                out[name] = p
        assert len(out) == len(params)
        return out

    @staticmethod
    def _from_v4(k: str) -> str:
        import re
        for pat, rep in _SEGFORMER_V4:
            k = re.sub(pat, rep, k)
        return k

    def load_state_dict(self, state_dict, strict: bool = True, **kwargs):
        return super().load_state_dict(OrderedDict((self._from_v4(k), v) for k, v in state_dict.items()), strict=strict, **kwargs)

    def custom_load(self, checkpoint):       # S/nets.py:351-356
        self.load_state_dict({str(k).replace("module.", ""): v for k, v in checkpoint["model_state_dict"].items()})
