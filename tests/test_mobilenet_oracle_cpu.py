"""Independent cross-check of oracle/mobilenet_oracle.py (PARITY UNPINNED: torchvision is absent from this image and the reference's
`MobileNet` wrapper fetches weights - S/nets.py:277-311).

The oracle is FUNCTIONAL code over a parameter dictionary.  Here the same published architecture is written a second time as an `nn.Module`
tree in the style of the library it comes from - torchvision.models.mobilenetv3 (`Conv2dNormActivation`, `SqueezeExcitation` with
`scale_activation=nn.Hardsigmoid`, `InvertedResidual` with its `block` Sequential and `use_res_connect`, `_mobilenet_v3_conf(dilated=True)`,
`_make_divisible`, BatchNorm2d(eps=0.001, momentum=0.01)) and torchvision.models.segmentation's `IntermediateLayerGetter` / `DeepLabHead` -
with `nn.Conv2d(groups=..., dilation=...)`, `nn.AdaptiveAvgPool2d`, `nn.Hardswish`, loaded through `load_state_dict(strict=True)` under the
library's own key names (`backbone.N.block.M...`, `classifier...`).  Logits (train and eval mode), the updated running statistics and every
parameter gradient must agree."""
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import mobilenet_oracle as M
from oracle import unet_oracle as O
from test_oracle_modules_cpu import ASPP


def _make_divisible(v, divisor=8, min_value=None):
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class Conv2dNormActivation(nn.Sequential):
    def __init__(self, cin, cout, kernel_size=3, stride=1, groups=1, dilation=1, norm_layer=nn.BatchNorm2d, activation_layer=nn.ReLU):
        padding = (kernel_size - 1) // 2 * dilation
        layers = [nn.Conv2d(cin, cout, kernel_size, stride, padding, dilation=dilation, groups=groups, bias=False), norm_layer(cout)]
        if activation_layer is not None:
            layers.append(activation_layer())
        super().__init__(*layers)
        self.out_channels = cout


class SqueezeExcitation(nn.Module):
    def __init__(self, channels, squeeze):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(channels, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, channels, 1)
        self.activation = nn.ReLU()
        self.scale_activation = nn.Hardsigmoid()

    def forward(self, x):
        scale = self.scale_activation(self.fc2(self.activation(self.fc1(self.avgpool(x)))))
        return scale * x


class InvertedResidualConfig:
    def __init__(self, cin, kernel, expanded, cout, use_se, activation, stride, dilation):
        self.input_channels, self.kernel, self.expanded_channels, self.out_channels = cin, kernel, expanded, cout
        self.use_se, self.use_hs, self.stride, self.dilation = use_se, activation == "HS", stride, dilation


class InvertedResidual(nn.Module):
    def __init__(self, cnf, norm_layer):
        super().__init__()
        self.use_res_connect = cnf.stride == 1 and cnf.input_channels == cnf.out_channels
        layers = []
        act = nn.Hardswish if cnf.use_hs else nn.ReLU
        if cnf.expanded_channels != cnf.input_channels:
            layers.append(Conv2dNormActivation(cnf.input_channels, cnf.expanded_channels, 1, norm_layer=norm_layer, activation_layer=act))
        stride = 1 if cnf.dilation > 1 else cnf.stride
        layers.append(Conv2dNormActivation(cnf.expanded_channels, cnf.expanded_channels, cnf.kernel, stride, groups=cnf.expanded_channels,
                                           dilation=cnf.dilation, norm_layer=norm_layer, activation_layer=act))
        if cnf.use_se:
            layers.append(SqueezeExcitation(cnf.expanded_channels, _make_divisible(cnf.expanded_channels // 4, 8)))
        layers.append(Conv2dNormActivation(cnf.expanded_channels, cnf.out_channels, 1, norm_layer=norm_layer, activation_layer=None))
        self.block = nn.Sequential(*layers)

    def forward(self, x):
        y = self.block(x)
        return y + x if self.use_res_connect else y


def mobilenet_v3_large_features(dilated=True):
    """torchvision `mobilenet_v3_large(dilated=dilated).features` (width_mult 1, reduced_tail False)"""
    dilation = 2 if dilated else 1
    c = InvertedResidualConfig
    setting = [c(16, 3, 16, 16, False, "RE", 1, 1), c(16, 3, 64, 24, False, "RE", 2, 1), c(24, 3, 72, 24, False, "RE", 1, 1),
               c(24, 5, 72, 40, True, "RE", 2, 1), c(40, 5, 120, 40, True, "RE", 1, 1), c(40, 5, 120, 40, True, "RE", 1, 1),
               c(40, 3, 240, 80, False, "HS", 2, 1), c(80, 3, 200, 80, False, "HS", 1, 1), c(80, 3, 184, 80, False, "HS", 1, 1),
               c(80, 3, 184, 80, False, "HS", 1, 1), c(80, 3, 480, 112, True, "HS", 1, 1), c(112, 3, 672, 112, True, "HS", 1, 1),
               c(112, 5, 672, 160, True, "HS", 2, dilation), c(160, 5, 960, 160, True, "HS", 1, dilation), c(160, 5, 960, 160, True, "HS", 1, dilation)]
    norm = partial(nn.BatchNorm2d, eps=0.001, momentum=0.01)
    layers = [Conv2dNormActivation(3, 16, 3, 2, norm_layer=norm, activation_layer=nn.Hardswish)]
    layers += [InvertedResidual(cnf, norm) for cnf in setting]
    layers.append(Conv2dNormActivation(160, 6 * 160, 1, norm_layer=norm, activation_layer=nn.Hardswish))
    return nn.Sequential(*layers)


class TVDeepLabV3MobileNet(nn.Module):
    """torchvision `deeplabv3_mobilenet_v3_large(num_classes=nc)` without the auxiliary head: the backbone's last feature -> DeepLabHead ->
    bilinear resize to the input size; forward returns d['out'] as the reference's wrapper does (S/nets.py:300-304)"""

    def __init__(self, nc):
        super().__init__()
        self.backbone = mobilenet_v3_large_features(dilated=True)
        self.classifier = nn.Sequential(ASPP(960, [12, 24, 36]), nn.Conv2d(256, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256), nn.ReLU(),
                                        nn.Conv2d(256, nc, 1))

    def forward(self, x):
        return F.interpolate(self.classifier(self.backbone(x)), size=x.shape[-2:], mode="bilinear", align_corners=False)


_KEYS = (("encoder.", "backbone."), ("aspp.convs.", "classifier.0.convs."), ("aspp.project.", "classifier.0.project."), ("head.0.", "classifier.1."),
         ("head.1.", "classifier.2."), ("classifier.", "classifier.4."))


def _to_torchvision(k):
    for mine, ref in _KEYS:
        if k.startswith(mine):
            return ref + k[len(mine):]
    return k


def test_make_divisible_and_the_parameter_table():
    assert [M.make_divisible(v // 4, 8) for v in (72, 120, 480, 672, 960)] == [_make_divisible(v // 4, 8) for v in (72, 120, 480, 672, 960)] == [24, 32, 120, 168, 240]
    net = TVDeepLabV3MobileNet(7)
    sd = {k: v for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}
    spec = {_to_torchvision(k): tuple(s) for k, s in M.param_spec(7)}
    assert {k: tuple(v.shape) for k, v in sd.items()} == spec
    # the same parameter count as the module tree at torchvision's default 21 classes
    n21 = sum(int(torch.tensor(s).prod()) for k, s in M.param_spec(21) if not M.is_buffer(k))
    assert n21 == sum(p.numel() for p in TVDeepLabV3MobileNet(21).parameters())


def test_oracle_matches_the_module_tree_forward_and_backward():
    NC, B, S = 6, 3, 96
    p = M.init_params(NC, seed=4)
    net = TVDeepLabV3MobileNet(NC)
    net.load_state_dict({_to_torchvision(k): v.clone() for k, v in p.items()}, strict=False)
    assert not [k for k in net.state_dict() if not k.endswith("num_batches_tracked") and k not in {_to_torchvision(q) for q in p}]
    img, lab = O.synthetic_tiles(B, S, NC, seed=2, structured=True)
    x = img.float()
    names = [k for k in p if not M.is_buffer(k)]
    for k in names:
        p[k].requires_grad_(True)
    net.train()
    want = net(x)
    got = M.forward(p, x, train=True)
    assert (got - want).abs().max() <= 1e-4 * want.abs().max()
    lw, lg = F.cross_entropy(want, lab.long(), ignore_index=0), O.cross_entropy(got, lab.long(), None, 0)
    assert abs(lw.item() - lg.item()) <= 1e-5 * abs(lw.item())
    lw.backward()
    grads = dict(zip(names, torch.autograd.grad(lg, [p[k] for k in names])))
    mods = dict(net.named_parameters())
    # (the projection BatchNorms' bias gradients vanish analytically - a BatchNorm behind the next 1x1 convolution removes a per-channel shift -
    #  and are f32 noise of ~1e-7 on both sides: a floor of 1e-4 of the largest gradient norm keeps them out of the relative figure)
    floor = 1e-4 * max(g.norm().item() for g in grads.values())
    worst = max((((grads[k] - mods[_to_torchvision(k)].grad).norm() / (mods[_to_torchvision(k)].grad.norm() + floor)).item(), k) for k in names)
    assert worst[0] <= 2e-3, worst
    sd = net.state_dict()
    for k in p:
        if M.is_buffer(k):      # running statistics after one train-mode forward (momentum 0.01 / eps 1e-3 in the backbone, torch defaults in the head)
            assert torch.allclose(p[k], sd[_to_torchvision(k)], rtol=1e-4, atol=1e-5), k
    net.eval()
    with torch.no_grad():
        want = net(x)
        got = M.forward({k: v.detach() for k, v in p.items()}, x, train=False)
    assert (got - want).abs().max() <= 1e-4 * want.abs().max()


def test_wrapper_state_dict_speaks_the_reference_key_names_on_cpu():
    """cvcs_amd.nets.DeepLabV3MobileNet without a device: `state_dict()` carries the names the reference's wrapper saves (`self.model =
    deeplabv3_mobilenet_v3_large(...)`, S/nets.py:288-293 -> `model.backbone.N...`, `model.classifier...`) and loads them back - with the
    DataParallel `module.` prefix the reference's `custom_load` strips (S/nets.py:306-311) and with the auxiliary head's tensors, which are kept
    and written out again; the module tree above loads the same dictionary strictly (minus the `model.` prefix)"""
    from cvcs_amd import nets
    net = nets.DeepLabV3MobileNet(7, "bf16")
    assert net.wrapper and net.returns_logits and not net.requires_context
    sd = net.state_dict()
    tv = TVDeepLabV3MobileNet(7)
    want = {"model." + k: tuple(v.shape) for k, v in tv.state_dict().items()}
    assert {k: tuple(v.shape) for k, v in sd.items()} == want
    tv.load_state_dict({k[len("model."):]: v for k, v in sd.items()}, strict=True)
    ck = {"module." + k: (torch.full_like(v, 0.5) if v.dtype == torch.float32 else v) for k, v in sd.items()}
    ck["module.model.aux_classifier.4.weight"] = torch.zeros(21, 10, 1, 1)
    other = nets.DeepLabV3MobileNet(7, "bf16")
    other.custom_load({"model_state_dict": ck})
    out = other.state_dict()
    assert all((v == 0.5).all() for k, v in out.items() if v.dtype == torch.float32 and "aux_classifier" not in k)
    assert (out["model.aux_classifier.4.weight"] == 0).all() and set(out) == set(sd) | {"model.aux_classifier.4.weight"}
