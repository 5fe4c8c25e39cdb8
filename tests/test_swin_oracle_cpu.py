"""The Swin-T + UPerNet oracle (oracle/swin_upernet_oracle.py) has no reference output to pin it (PARITY UNPINNED, see its
header).  Cross-check on the CPU against an independent implementation of the same published architectures:
`transformers.UperNetForSemanticSegmentation` over a `SwinBackbone`, built from a LOCAL config (no fetch), same weights - logits
and every backbone stage, eval and train mode, 64x64 (maps smaller than a window: all padding) and 224x224 (no padding in stage 1,
shifted windows with real regions)."""
import pytest
import torch

from oracle import swin_upernet_oracle as W

transformers = pytest.importorskip("transformers")


def _hf(num_classes):
    from transformers import SwinConfig, UperNetConfig, UperNetForSemanticSegmentation
    bc = SwinConfig(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7, drop_path_rate=0.0,
                    out_features=["stage1", "stage2", "stage3", "stage4"])
    return UperNetForSemanticSegmentation(UperNetConfig(backbone_config=bc, hidden_size=512, num_labels=num_classes, use_auxiliary_head=False))


@pytest.mark.parametrize("S,train", [(64, False), (64, True), (224, False)])
def test_oracle_matches_transformers_upernet_swin(S, train):
    NC = 5
    p = W.init_params(NC, seed=2)
    g = torch.Generator().manual_seed(1)
    for k in p:
        if k.endswith("running_mean"):
            p[k] = torch.randn(p[k].shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            p[k] = 0.5 + torch.rand(p[k].shape, generator=g)
    m = _hf(NC)
    missing, unexpected = m.load_state_dict({k: v.clone() for k, v in p.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith("num_batches_tracked") or k.startswith("backbone.swin.layernorm") for k in missing), missing
    m.train(train)
    x = torch.randint(0, 256, (2, 3, S, S), generator=g).float()
    with torch.no_grad():
        want = m(x).logits
        hf_feats = m.backbone(x).feature_maps
        got, feats = W.forward({k: v.clone() for k, v in p.items()}, x, train=train, return_feats=True)
    for a, b in zip(hf_feats, feats):
        assert a.shape == b.shape and (a - b).abs().max().item() <= 2e-4 * max(1.0, a.abs().max().item())
    assert got.shape == want.shape == (2, NC, S, S)
    assert (got - want).abs().max().item() <= 2e-4 * max(1.0, want.abs().max().item())


def test_parameter_count_and_flops():
    n = sum(int(torch.tensor(s).prod()) for k, s in W.param_spec(150) if not W.is_buffer(k))
    # Swin-T backbone 27.5 M + UPerNet head (hidden 512): HF's count for the same config minus the unused final LayerNorm
    m = _hf(150)
    want = sum(v.numel() for k, v in m.named_parameters() if not k.startswith("backbone.swin.layernorm"))
    assert n == want, (n, want)
    assert 400 < W.forward_gflop_per_tile(16, 512) < 600
