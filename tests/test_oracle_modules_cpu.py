"""Independent cross-check of the UNPINNED oracles' halves that nothing else checked: the ResNet-UNet decoder and the
DeepLab ASPP / image pooling / heads / V3+ decoder, and the dilated-encoder rule.

The oracles (oracle/resnet_unet_oracle.py, oracle/deeplab_oracle.py) are FUNCTIONAL code over a parameter dictionary.  Here the same
published architectures are written a second time as `nn.Module` trees in the style of the libraries they come from - torchvision's
`ResNet._make_layer(dilate=...)` with its `previous_dilation` book-keeping, `ASPPConv` / `ASPPPooling` / `ASPP` / `DeepLabHead`
(torchvision.models.segmentation.deeplabv3), an `nn.Upsample` + `Conv2dReLU` decoder block (segmentation_models_pytorch's UnetDecoder
shape) - with `nn.Conv2d(dilation=...)`, `nn.AdaptiveAvgPool2d`, `nn.BatchNorm2d`, `nn.Sequential`, loaded through `load_state_dict(strict=True)`
under the libraries' own key names.  Logits (train and eval mode) and every parameter gradient must agree."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import deeplab_oracle as D
from oracle import resnet_unet_oracle as R
from oracle import unet_oracle as O


# ------------------------------------------------------------------------------------------------ torchvision-style ResNet
class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)   # v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class TVResNet(nn.Module):
    """torchvision.models.resnet.ResNet without avgpool / fc; forward returns the five feature maps"""

    def __init__(self, block, layers, replace_stride_with_dilation=(False, False, False)):
        super().__init__()
        self.inplanes, self.dilation = 64, 1
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, dilate=replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2, dilate=replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2, dilate=replace_stride_with_dilation[2])

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        downsample, previous_dilation = None, self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, previous_dilation)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, dilation=self.dilation))
        return nn.Sequential(*layers)

    def forward(self, x):
        f1 = self.relu(self.bn1(self.conv1(x)))
        f2 = self.layer1(self.maxpool(f1))
        f3 = self.layer2(f2)
        f4 = self.layer3(f3)
        f5 = self.layer4(f4)
        return [f1, f2, f3, f4, f5]


# ------------------------------------------------------------------------------------------------ torchvision-style DeepLabV3 head
class ASPPConv(nn.Sequential):
    def __init__(self, cin, cout, dilation):
        super().__init__(nn.Conv2d(cin, cout, 3, padding=dilation, dilation=dilation, bias=False), nn.BatchNorm2d(cout), nn.ReLU())


class ASPPPooling(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(nn.AdaptiveAvgPool2d(1), nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU())

    def forward(self, x):
        size = x.shape[-2:]
        for mod in self:
            x = mod(x)
        return F.interpolate(x, size=size, mode="bilinear", align_corners=False)


class ASPP(nn.Module):
    def __init__(self, cin, rates, cout=256):
        super().__init__()
        mods = [nn.Sequential(nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU())]
        mods += [ASPPConv(cin, cout, r) for r in rates]
        mods.append(ASPPPooling(cin, cout))
        self.convs = nn.ModuleList(mods)
        # (torchvision appends nn.Dropout(0.5) here: stochastic in train mode, identity in eval - left out on both sides)
        self.project = nn.Sequential(nn.Conv2d(len(mods) * cout, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU())

    def forward(self, x):
        return self.project(torch.cat([c(x) for c in self.convs], dim=1))


class TVDeepLabV3(nn.Module):
    """torchvision.models.segmentation.deeplabv3_resnet*: IntermediateLayerGetter backbone + DeepLabHead, output resized to the input"""

    def __init__(self, block, layers, nc, rswd, rates):
        super().__init__()
        self.backbone = TVResNet(block, layers, rswd)
        self.classifier = nn.Sequential(ASPP(512 * block.expansion, rates), nn.Conv2d(256, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256),
                                        nn.ReLU(), nn.Conv2d(256, nc, 1))

    def forward(self, x):
        out = self.classifier(self.backbone(x)[4])
        return F.interpolate(out, size=x.shape[-2:], mode="bilinear", align_corners=False)


class DeepLabV3PlusModule(nn.Module):
    """Chen et al. 2018: ASPP on the stride-16 feature, 1x1 reduction of the stride-4 feature, concat, two 3x3 convs, classifier, x4"""

    def __init__(self, nc, low_channels=D.LOW_LEVEL_CHANNELS):
        super().__init__()
        self.backbone = TVResNet(Bottleneck, [3, 4, 6, 3], (False, False, True))
        self.aspp = ASPP(2048, (6, 12, 18))
        self.low = nn.Sequential(nn.Conv2d(256, low_channels, 1, bias=False), nn.BatchNorm2d(low_channels), nn.ReLU())
        self.refine = nn.Sequential(nn.Conv2d(256 + low_channels, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256), nn.ReLU(),
                                    nn.Conv2d(256, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256), nn.ReLU())
        self.cls = nn.Conv2d(256, nc, 1)

    def forward(self, x):
        f = self.backbone(x)
        a = F.interpolate(self.aspp(f[4]), size=f[1].shape[-2:], mode="bilinear", align_corners=False)
        z = self.cls(self.refine(torch.cat([a, self.low(f[1])], dim=1)))
        return F.interpolate(z, size=x.shape[-2:], mode="bilinear", align_corners=False)


# ------------------------------------------------------------------------------------------------ U-Net decoder (smp-style)
class Conv2dReLU(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class DecoderBlock(nn.Module):
    def __init__(self, cin, cskip, cout):
        super().__init__()
        self.up = nn.Upsample(scale_factor=2, mode="bilinear")     # the reference's own up-sampling op (S/blocks.py:29)
        self.conv1 = Conv2dReLU(cin + cskip, cout)
        self.conv2 = Conv2dReLU(cout, cout)

    def forward(self, x, skip=None):
        x = self.up(x)
        if skip is not None:
            x = torch.cat([x, skip], dim=1)
        return self.conv2(self.conv1(x))


class ResNetUNetModule(nn.Module):
    def __init__(self, block, layers, nc, dec=R.DECODER_CHANNELS):      # (256, 128, 64, 32, 16)
        super().__init__()
        self.encoder = TVResNet(block, layers)
        w = [64 * block.expansion * m for m in (1, 2, 4, 8)]
        skips = [w[2], w[1], w[0], 64, 0]
        cin, blocks = w[3], []
        for c, s in zip(dec, skips):
            blocks.append(DecoderBlock(cin, s, c))
            cin = c
        self.blocks = nn.ModuleList(blocks)
        self.head = nn.Conv2d(cin, nc, 1)

    def forward(self, x):
        f = self.encoder(x)
        skips = [f[3], f[2], f[1], f[0], None]
        h = f[4]
        for b, s in zip(self.blocks, skips):
            h = b(h, s)
        return self.head(h)


# ------------------------------------------------------------------------------------------------ key maps (written here, independently)
def _map_unet(k):
    if k.startswith("decoder.blocks."):
        _, _, i, conv, j, leaf = k.split(".")
        return f"blocks.{i}.{conv}.{j}.{leaf}"
    if k.startswith("segmentation_head.0."):
        return "head." + k.split(".")[-1]
    return k      # encoder.* keeps torchvision's names


def _map_deeplabv3(k):
    if k.startswith("encoder."):
        return "backbone." + k[len("encoder."):]
    if k.startswith("aspp."):
        return "classifier.0." + k[len("aspp."):]
    if k.startswith("head.0."):
        return "classifier.1." + k.split(".")[-1]
    if k.startswith("head.1."):
        return "classifier.2." + k.split(".")[-1]
    return "classifier.4." + k.split(".")[-1]


def _map_v3plus(k):
    if k.startswith("encoder."):
        return "backbone." + k[len("encoder."):]
    if k.startswith("low_level."):
        return "low." + k[len("low_level."):]
    if k.startswith("decoder.conv1."):
        return "refine." + {"0": "0", "1": "1"}[k.split(".")[2]] + "." + k.split(".")[-1]
    if k.startswith("decoder.conv2."):
        return "refine." + {"0": "3", "1": "4"}[k.split(".")[2]] + "." + k.split(".")[-1]
    if k.startswith("classifier."):
        return "cls." + k.split(".")[-1]
    return k      # aspp.*


def _load(module, p, keymap):
    sd = {keymap(k): v.clone() for k, v in p.items()}
    for name, buf in module.named_buffers():
        if name.endswith("num_batches_tracked"):
            sd[name] = buf.clone()
    module.load_state_dict(sd, strict=True)
    return {keymap(k): k for k in p}


def _compare(module, keymap, p, oracle_fwd, x, lab, tol=2e-4):
    inv = _load(module, p, keymap)
    module.train()
    z = module(x)
    loss = O.cross_entropy(z, lab.long(), None, 0)
    loss.backward()
    po = {k: (v.clone().requires_grad_(True) if not R.is_buffer(k) else v.clone()) for k, v in p.items()}
    zo = oracle_fwd(po, x, True)
    lo = O.cross_entropy(zo, lab.long(), None, 0)
    names = [k for k in po if not R.is_buffer(k)]
    go = dict(zip(names, torch.autograd.grad(lo, [po[k] for k in names])))
    scale = zo.abs().max().item()
    assert (z - zo).abs().max().item() <= tol * scale, "train-mode logits"
    assert abs(loss.item() - lo.item()) <= 1e-5 * max(1.0, abs(lo.item()))
    worst = 0.0
    for name, prm in module.named_parameters():
        g, ref = prm.grad, go[inv[name]]
        e = (g - ref).norm().item() / max(ref.norm().item(), 1e-12)
        worst = max(worst, e)
        assert e <= 2e-3, (name, e)
    # running statistics after the train-mode pass, then eval mode
    for name, buf in module.named_buffers():
        if not name.endswith("num_batches_tracked"):
            assert torch.allclose(buf, po[inv[name]], rtol=1e-4, atol=1e-5), name
    module.eval()
    with torch.no_grad():
        ze, zoe = module(x), oracle_fwd(po, x, False)
    assert (ze - zoe).abs().max().item() <= tol * zoe.abs().max().item(), "eval-mode logits"
    return worst


def test_resnet_unet_oracle_matches_a_module_definition():
    for arch, block, layers in (("resnet18", BasicBlock, [2, 2, 2, 2]), ("resnet50", Bottleneck, [3, 4, 6, 3])):
        NC = 5
        p = R.init_params(arch, NC, seed=5)
        img, lab = O.synthetic_tiles(2, 64, NC, seed=3, structured=True)
        worst = _compare(ResNetUNetModule(block, layers, NC), _map_unet, p, lambda q, x, tr: R.forward(q, x, arch, train=tr), img.float(), lab)
        print(f"{arch}-UNet: module vs oracle, worst parameter-gradient difference {worst:.2e}")


def test_deeplabv3_oracle_matches_a_torchvision_style_module():
    """the reference's `Resnet101` family (S/nets.py:234-257): DeepLabV3 at output stride 8 (replace_stride_with_dilation [False, True, True]),
    ASPP 12/24/36, DeepLabHead - on ResNet-50 depth to keep the CPU test short (the dilation rule is depth-independent)"""
    NC = 6
    p = D.init_params("resnet50", NC, seed=5, plus=False)
    img, lab = O.synthetic_tiles(2, 64, NC, seed=4, structured=True)
    m = TVDeepLabV3(Bottleneck, [3, 4, 6, 3], NC, (False, True, True), (12, 24, 36))
    worst = _compare(m, _map_deeplabv3, p, lambda q, x, tr: D.forward(q, x, "resnet50", train=tr, output_stride=8, plus=False), img.float(), lab)
    print(f"DeepLabV3 (OS 8): module vs oracle, worst parameter-gradient difference {worst:.2e}")


def test_deeplabv3plus_oracle_matches_a_module_definition():
    NC = 6
    p = D.init_params("resnet50", NC, seed=6, plus=True)
    img, lab = O.synthetic_tiles(2, 64, NC, seed=5, structured=True)
    worst = _compare(DeepLabV3PlusModule(NC), _map_v3plus, p, lambda q, x, tr: D.forward(q, x, "resnet50", train=tr, output_stride=16, plus=True),
                     img.float(), lab)
    print(f"DeepLabV3+ (OS 16, ASPP 6/12/18): module vs oracle, worst parameter-gradient difference {worst:.2e}")
