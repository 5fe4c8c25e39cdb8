"""The ResNet-UNet oracle (oracle/resnet_unet_oracle.py) has no reference output to pin it (the reference has no such
model: PARITY UNPINNED, see the oracle's header).  What CAN be checked on the CPU:
  * its encoder against an independent implementation of the same published architecture - transformers.ResNetModel
    built from a LOCAL config (no fetch), same weights, eval and train mode;
  * parameter counts against the published ResNet-18/50 backbone sizes;
  * its decoder against plain nn.Module code written the other way round (nn.Sequential modules);
  * the bf16-emulation mode: identity on bf16-representable data paths, close to fp32 otherwise.
"""
import pytest
import torch
import torch.nn as nn

from oracle import resnet_unet_oracle as R

transformers = pytest.importorskip("transformers")


def _hf_resnet(arch):
    from transformers import ResNetConfig, ResNetModel
    if arch == "resnet50":
        cfg = ResNetConfig()   # bottleneck, depths [3,4,6,3], hidden [256,512,1024,2048], stride on the 3x3 (v1.5)
    else:
        cfg = ResNetConfig(layer_type="basic", depths=[2, 2, 2, 2], hidden_sizes=[64, 128, 256, 512])
    return ResNetModel(cfg)


def _load_into_hf(model, p, arch):
    kind, depths, _ = R.ARCHS[arch]
    sd = {}

    def put(dst, src_conv, src_bn):
        sd[dst + ".convolution.weight"] = p[src_conv + ".weight"]
        for a in ("weight", "bias", "running_mean", "running_var"):
            sd[dst + ".normalization." + a] = p[src_bn + "." + a]

    put("embedder.embedder", "encoder.conv1", "encoder.bn1")
    nconv = 3 if kind == "bottleneck" else 2
    for s, n in enumerate(depths):
        for b in range(n):
            src = f"encoder.layer{s + 1}.{b}"
            dst = f"encoder.stages.{s}.layers.{b}"
            for j in range(nconv):
                put(f"{dst}.layer.{j}", f"{src}.conv{j + 1}", f"{src}.bn{j + 1}")
            if (src + ".downsample.0.weight") in p:
                put(f"{dst}.shortcut", src + ".downsample.0", src + ".downsample.1")
    missing, unexpected = model.load_state_dict({k: v.detach().clone() for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
@pytest.mark.parametrize("train", [False, True])
def test_encoder_matches_transformers_resnet(arch, train):
    torch.manual_seed(0)
    p = R.init_params(arch, 5, seed=3)
    # non-trivial running statistics so that eval mode exercises them
    g = torch.Generator().manual_seed(5)
    for k in p:
        if k.endswith("running_mean"):
            p[k] = torch.randn(p[k].shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            p[k] = 0.5 + torch.rand(p[k].shape, generator=g)
    m = _hf_resnet(arch)
    _load_into_hf(m, p, arch)
    m.train(train)
    x = torch.randint(0, 256, (2, 3, 64, 64), generator=g).float()
    with torch.no_grad():
        hf = m(x, output_hidden_states=True).hidden_states   # embedder (after pool), stage 1..4
        mine = R.encoder_features({k: v.clone() for k, v in p.items()}, x, arch, train=train)
    for a, b in zip(hf[1:], mine[1:]):
        assert a.shape == b.shape
        assert (a - b).abs().max().item() <= 1e-4 * max(1.0, a.abs().max().item())


def test_parameter_counts_are_the_published_backbones():
    """ResNet-18 / ResNet-50 backbones (no fc): 11 176 512 / 23 508 032 parameters (torchvision's numbers)"""
    for arch, want in (("resnet18", 11176512), ("resnet50", 23508032)):
        n = sum(int(torch.tensor(s).prod()) for k, s in R.param_spec(arch, 5)
                if k.startswith("encoder.") and not R.is_buffer(k))
        assert n == want, (arch, n)


class _Dec(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU())
        self.conv2 = nn.Sequential(nn.Conv2d(cout, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU())


def test_decoder_and_head_match_module_code():
    arch = "resnet18"
    p = R.init_params(arch, 4, seed=1)
    x = torch.randint(0, 256, (2, 3, 64, 64), generator=torch.Generator().manual_seed(2)).float()
    p2 = {k: v.clone() for k, v in p.items()}
    logits, acts = R.forward(p2, x, arch, train=True, return_acts=True)
    feats = acts["features"]
    skips = [feats[3], feats[2], feats[1], feats[0], None]
    h = feats[4]
    up = nn.Upsample(scale_factor=2, mode="bilinear")    # the reference's up-sampling op, S/blocks.py:29
    for i in range(5):
        h = up(h)
        if skips[i] is not None:
            h = torch.cat([h, skips[i]], 1)
        w = p[f"decoder.blocks.{i}.conv1.0.weight"]
        blk = _Dec(w.shape[1], w.shape[0])
        sd = {}
        for j in (1, 2):
            sd[f"conv{j}.0.weight"] = p[f"decoder.blocks.{i}.conv{j}.0.weight"]
            for a in ("weight", "bias", "running_mean", "running_var"):
                sd[f"conv{j}.1.{a}"] = p[f"decoder.blocks.{i}.conv{j}.1.{a}"]
        blk.load_state_dict(sd, strict=False)
        blk.train()
        h = blk.conv2(blk.conv1(h))
    head = nn.Conv2d(p["segmentation_head.0.weight"].shape[1], 4, 1)
    head.load_state_dict({"weight": p["segmentation_head.0.weight"], "bias": p["segmentation_head.0.bias"]})
    ref = head(h)
    assert (ref - logits).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())


def test_layer_table_matches_parameter_shapes_and_known_flops():
    for arch in ("resnet18", "resnet50"):
        spec = dict(R.param_spec(arch, 16))
        for name, ci, co, k, _, _ in R.conv_table(arch, 16, 512):
            assert spec[name + ".weight"] == (co, ci, k, k)
    # ResNet-50 backbone at 224x224 is the textbook 4.09 GMAC (convs only: 4.087 GMAC)
    enc = sum(2.0 * ci * co * k * k * ho * ho for n, ci, co, k, _, ho in R.conv_table("resnet50", 16, 224)
              if n.startswith("encoder."))
    assert abs(enc / 2e9 - 4.087) < 0.01, enc / 2e9


def test_bf16_emulation_mode():
    arch = "resnet18"
    p = R.init_params(arch, 5, seed=0)
    x = torch.randint(0, 256, (2, 3, 64, 64), generator=torch.Generator().manual_seed(0)).float()
    a = R.forward({k: v.clone() for k, v in p.items()}, x, arch, train=True)
    b = R.forward({k: v.clone() for k, v in p.items()}, x, arch, train=True, emulate_bf16=True)
    rel = ((a - b).abs().max() / a.abs().max()).item()
    assert 0 < rel < 0.15, rel        # bf16 storage is visible, and bounded
    # gradients flow through the rounding (straight-through) and the trainer steps
    tr = R.OracleTrainer(arch, 5, "SGD2", ignore_index=0, seed=0, emulate_bf16=True)
    lab = torch.randint(0, 5, (2, 64, 64), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
    l0, _, g = tr.step(x.to(torch.uint8), lab)
    l1, _, _ = tr.step(x.to(torch.uint8), lab)
    assert l1 < l0 and all(torch.isfinite(v).all() for v in g.values())
