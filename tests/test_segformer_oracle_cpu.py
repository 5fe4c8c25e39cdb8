"""CPU-only: oracle/segformer_oracle.py PINNED against the third-party implementation the reference wraps
(`transformers.SegformerForSemanticSegmentation(SegformerConfig())`, built from a local config - no fetch), assembled the way
/root/reference/source/scripts/nets.py:313-349 assembles `SegformerMod`."""
import pytest
import torch
from torch import nn

transformers = pytest.importorskip("transformers")

from oracle import segformer_oracle as SO   # noqa: E402
from oracle import unet_oracle as U        # noqa: E402

NC = 5


class _Wrapped(nn.Module):
    """what nets.py:313-349 builds with pretrained=False: the library model with an NC-class 1x1 classifier, then the reference's
    ConvTranspose / ConvTranspose / Conv tail; the input normalised with the ImageNet constants on the raw 0..255 scale"""

    def __init__(self, nc):
        super().__init__()
        from transformers import SegformerConfig, SegformerForSemanticSegmentation
        self.segformer = SegformerForSemanticSegmentation(SegformerConfig())
        cin = self.segformer.decode_head.classifier.in_channels
        self.segformer.decode_head.classifier = nn.Conv2d(cin, nc, kernel_size=(1, 1), stride=(1, 1))
        self.seq = nn.Sequential(nn.ConvTranspose2d(nc, nc, 8, stride=2, padding=3), nn.ReLU(), nn.ConvTranspose2d(nc, nc, 4, stride=2, padding=1),
                                 nn.ReLU(), nn.Conv2d(nc, nc, kernel_size=3, padding=1))

    def forward(self, x):
        mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        x = (x.to(torch.float32) - mean) / std
        return self.seq(self.segformer(x).logits)


@pytest.fixture(scope="module")
def pair():
    torch.manual_seed(0)
    m = _Wrapped(NC)
    p = SO.init_params(NC, seed=3)
    sd = m.state_dict()
    mine = {k for k, _ in SO.param_spec(NC)}
    theirs = {k for k in sd if not k.endswith("num_batches_tracked")}
    assert mine == theirs, (sorted(mine - theirs)[:5], sorted(theirs - mine)[:5])
    for k, shape in SO.param_spec(NC):
        assert tuple(sd[k].shape) == tuple(shape), k
    m.load_state_dict({**{k: v.clone() for k, v in p.items()}, **{k: v for k, v in sd.items() if k.endswith("num_batches_tracked")}}, strict=True)
    return m, p


def _input(B=2, S=64, seed=1):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (B, 3, S, S), generator=g, dtype=torch.uint8)


def test_names_shapes_and_the_layer_norm_epsilon_of_the_built_modules(pair):
    m, _ = pair
    lns = [mod for mod in m.modules() if isinstance(mod, nn.LayerNorm)]
    assert lns and all(ln.eps == SO.CONFIGS["b0"]["ln_eps"] for ln in lns)
    bn = m.segformer.decode_head.batch_norm
    assert bn.eps == SO.BN_EPS and bn.momentum == SO.BN_MOMENTUM
    rates = [b.drop_path.drop_prob if hasattr(b.drop_path, "drop_prob") else 0.0 for s in m.segformer.segformer.stages for b in s.blocks]
    assert rates == pytest.approx(SO.drop_path_rates(SO.CONFIGS["b0"]))
    assert m.segformer.decode_head.dropout.p == SO.CONFIGS["b0"]["classifier_dropout"]


def test_eval_logits_and_stage_outputs_match_the_library(pair):
    m, p = pair
    m.eval()
    x = _input()
    with torch.no_grad():
        ref = m(x)
        hs = m.segformer.segformer(SO.preprocess(x), output_hidden_states=True).hidden_states
        mine, feats, z = SO.forward({k: v.clone() for k, v in p.items()}, x, train=False, return_parts=True)
    assert ref.shape == (2, NC, 64, 64)
    for a, b in zip(hs, feats):
        assert torch.allclose(a, b, atol=2e-4, rtol=1e-4), float((a - b).abs().max())
    assert float((ref - mine).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max()))


def test_train_step_under_one_torch_seed_matches_the_library_logits_gradients_running_statistics(pair):
    """train mode: DropPath (rates 0 ... 0.1) and the head's Dropout(0.1) draw from the torch generator - the oracle makes the same
    calls in the same order, so one seed gives one step.  B = 8 so that some DropPath draws actually drop a sample."""
    m, p0 = pair
    p = {k: v.clone() for k, v in p0.items()}
    m.load_state_dict({**p, **{k: v for k, v in m.state_dict().items() if k.endswith("num_batches_tracked")}}, strict=True)
    m.train()
    x = _input(B=8, S=64, seed=5)
    y = torch.randint(0, NC, (8, 64, 64), generator=torch.Generator().manual_seed(6))
    torch.manual_seed(1234)
    ref = m(x)
    loss_ref = nn.functional.cross_entropy(ref, y)
    m.zero_grad()
    loss_ref.backward()
    names = [k for k in p if not SO.is_buffer(k)]
    for k in names:
        p[k].requires_grad_(True)
    torch.manual_seed(1234)
    mine = SO.forward(p, x, train=True, rng="torch")
    loss = U.cross_entropy(mine, y, None, -100)
    grads = dict(zip(names, torch.autograd.grad(loss, [p[k] for k in names])))
    assert float((ref - mine).detach().abs().max()) <= 2e-4 * max(1.0, float(ref.detach().abs().max()))
    assert abs(float(loss) - float(loss_ref)) <= 1e-5
    mp = dict(m.named_parameters())
    worst = 0.0
    for k in names:
        g, r = grads[k], mp[k].grad
        # (the biases in front of the head's BatchNorm and the key biases have mathematically zero gradients: 1e-8 of rounding noise)
        worst = max(worst, float((g - r).abs().max()) / max(float(r.abs().max()), 1e-5))
    assert worst <= 1e-3, worst
    sd = m.state_dict()
    for k in ("segformer.decode_head.batch_norm.running_mean", "segformer.decode_head.batch_norm.running_var"):
        assert torch.allclose(sd[k], p[k], atol=1e-5, rtol=1e-5), k
    # the draw was not trivial: at least one DropPath call dropped a sample
    torch.manual_seed(1234)
    drops = [float(torch.floor(torch.rand((8, 1, 1)) + (1 - r)).min()) for r in SO.drop_path_rates(SO.CONFIGS["b0"])[1:] for _ in (0, 1)]
    assert min(drops) == 0.0


def test_explicit_masks_reproduce_the_torch_stream(pair):
    """`masks=` (what the HIP path replays from its own generator) is the same arithmetic as `rng="torch"` given the same draws"""
    _, p0 = pair
    p = {k: v.clone() for k, v in p0.items()}
    x = _input(B=4, S=64, seed=7)
    cfg = SO.CONFIGS["b0"]
    seen = {}
    real = SO.F.dropout

    def spy(h, p_drop, training=True):       # the head's Bernoulli draw, as a keep / (1 - p) mask (where h == 0 the draw does not matter)
        out = real(h, p_drop, training=training)
        seen["keep"] = torch.where(h != 0, out / torch.where(h != 0, h, torch.ones_like(h)), torch.ones_like(h))
        return out
    SO.F.dropout = spy
    try:
        torch.manual_seed(99)
        a = SO.forward({k: v.clone() for k, v in p.items()}, x, train=True, rng="torch")
    finally:
        SO.F.dropout = real
    torch.manual_seed(99)
    dp = []
    for r in SO.drop_path_rates(cfg):
        for _ in (0, 1):
            if r > 0:
                dp.append(torch.floor(torch.rand((4, 1, 1)) + (1 - r)).view(-1) / (1 - r))
    keep = seen["keep"]
    assert 0.05 < float((keep == 0).float().mean()) < 0.15
    b = SO.forward({k: v.clone() for k, v in p.items()}, x, train=True, masks=dict(drop_path=dp, dropout=keep))
    assert torch.allclose(a, b, atol=1e-4, rtol=1e-4)       # (keep = out / h re-multiplied: an ulp per element)


def test_v4_checkpoint_names_map_onto_the_v5_names():
    """a real CVCS checkpoint was written under the transformers release of its day (4.x module layout)"""
    v4 = ["segformer.segformer.encoder.patch_embeddings.0.proj.weight", "segformer.segformer.encoder.patch_embeddings.2.layer_norm.bias",
          "segformer.segformer.encoder.block.1.0.layer_norm_1.weight", "segformer.segformer.encoder.block.0.1.attention.self.query.bias",
          "segformer.segformer.encoder.block.0.1.attention.self.key.weight", "segformer.segformer.encoder.block.2.1.attention.self.value.weight",
          "segformer.segformer.encoder.block.0.0.attention.self.sr.weight", "segformer.segformer.encoder.block.0.0.attention.self.layer_norm.bias",
          "segformer.segformer.encoder.block.3.1.attention.output.dense.weight", "segformer.segformer.encoder.block.3.1.layer_norm_2.bias",
          "segformer.segformer.encoder.block.3.1.mlp.dense1.weight", "segformer.segformer.encoder.block.3.1.mlp.dwconv.dwconv.bias",
          "segformer.segformer.encoder.block.3.1.mlp.dense2.bias", "segformer.segformer.encoder.layer_norm.3.weight",
          "segformer.decode_head.linear_c.2.proj.weight", "segformer.decode_head.linear_fuse.weight", "segformer.decode_head.batch_norm.running_var",
          "segformer.decode_head.classifier.bias", "seq.0.weight", "seq.4.bias"]
    names = {k for k, _ in SO.param_spec(NC)}
    mapped = [SO.from_v4_name(k) for k in v4]
    assert all(k in names for k in mapped), [k for k in mapped if k not in names]
    assert len(set(mapped)) == len(v4)
    assert all(SO.from_v4_name(k) == k for k in names)


def test_flop_count_against_a_hook_count():
    """forward_gflop_per_tile vs multiply-adds counted by module hooks on the library model + the tail"""
    m = _Wrapped(NC).eval()
    total = [0.0]

    def hook(mod, inp, out):
        if isinstance(mod, nn.Linear):
            total[0] += 2.0 * out.numel() * mod.in_features
        elif isinstance(mod, nn.Conv2d):
            total[0] += 2.0 * out.numel() * mod.in_channels // mod.groups * mod.kernel_size[0] * mod.kernel_size[1]
        elif isinstance(mod, nn.ConvTranspose2d):
            total[0] += 2.0 * out.numel() * mod.in_channels * (mod.kernel_size[0] // 2) ** 2

    for mod in m.modules():
        if isinstance(mod, (nn.Linear, nn.Conv2d, nn.ConvTranspose2d)):
            mod.register_forward_hook(hook)
    S = 64
    with torch.no_grad():
        m(_input(B=1, S=S))
    cfg = SO.CONFIGS["b0"]
    attn, H = 0.0, S
    for C, depth, sr, stride in zip(cfg["hidden"], cfg["depths"], cfg["sr"], cfg["strides"]):
        H //= stride
        attn += depth * 2.0 * 2 * (H * H) * (H * H // (sr * sr)) * C
    assert SO.forward_gflop_per_tile(NC, S) * 1e9 == pytest.approx(total[0] + attn, rel=1e-6)
