"""GPU parity of the Swin-T + UPerNet path (BASELINE.json configs[3]) - building-block kernels against ATen / the oracle's own
pieces, the network against `oracle/swin_upernet_oracle.py` (PARITY UNPINNED: no reference model; the oracle is cross-checked
against transformers.UperNetForSemanticSegmentation on the CPU, tests/test_swin_oracle_cpu.py)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cvcs_amd import ops  # noqa: E402
from cvcs_amd.ops import View  # noqa: E402
from oracle import swin_upernet_oracle as W  # noqa: E402

DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16]


def to_nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV).contiguous()


def from_nhwc(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def rq(x, dtype):
    return x.to(dtype).float()


def close(got, ref, rel, what=""):
    scale = max(1e-6, ref.abs().max().item())
    err = (got - ref).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs {rel:.1e} * {scale:.3e}"


def tok(t, dtype):
    """CPU [M, C] -> device token view [1, M, 1, C]"""
    return ops.view(t.to(dtype).to(DEV).contiguous().view(1, t.shape[0], 1, t.shape[1]))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C_,M", [(96, 70), (192, 45), (384, 33), (1536, 9), (768, 130), (2048, 7)])
def test_layernorm_forward_and_backward(dtype, C_, M):
    g = torch.Generator().manual_seed(C_)
    x = rq(torch.randn(M, C_, generator=g) * 2 + 0.5, dtype).requires_grad_(True)
    gamma = (torch.rand(C_, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C_, generator=g) * 0.2).requires_grad_(True)
    ref = F.layer_norm(x, (C_,), gamma, beta, 1e-5)
    go = rq(torch.randn(M, C_, generator=g), dtype)
    ref.backward(go)
    xv, gv = tok(x.detach(), dtype), tok(go, dtype)
    out, dx = torch.empty_like(xv.t), torch.empty_like(xv.t)
    mean, invstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.layernorm_fwd(xv, gamma.detach().to(DEV), beta.detach().to(DEV), ops.view(out), mean, invstd)
    rows = ops.layernorm_rows(M)
    part = torch.zeros(rows * 2 * C_, device=DEV)
    ops.layernorm_bwd(xv, gv, gamma.detach().to(DEV), mean, invstd, ops.view(dx), part)
    gb = torch.empty(2 * C_, device=DEV)
    ops.colsum_finalize(part, rows, 2 * C_, gb)
    torch.cuda.synchronize()
    t = 2e-5 if dtype == torch.float32 else 1e-2
    close(out.float().cpu().view(M, C_), ref.detach(), t, "LayerNorm forward")
    close(dx.float().cpu().view(M, C_), x.grad, 5e-5 if dtype == torch.float32 else 1e-2, "LayerNorm dx")
    close(gb[:C_].cpu(), gamma.grad, 1e-4, "dgamma")
    close(gb[C_:].cpu(), beta.grad, 1e-4, "dbeta")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gelu_forward_and_backward(dtype):
    g = torch.Generator().manual_seed(0)
    x = rq(torch.randn(50, 384, generator=g) * 2, dtype).requires_grad_(True)
    ref = F.gelu(x)
    go = rq(torch.randn(50, 384, generator=g), dtype)
    ref.backward(go)
    xv = tok(x.detach(), dtype)
    out, dx = torch.empty_like(xv.t), torch.empty_like(xv.t)
    ops.gelu(xv, ops.view(out))
    ops.gelu(xv, ops.view(dx), g=tok(go, dtype))
    torch.cuda.synchronize()
    t = 1e-6 if dtype == torch.float32 else 1e-2
    close(out.float().cpu().view(50, 384), ref.detach(), max(t, 2e-6), "GELU")
    close(dx.float().cpu().view(50, 384), x.grad, max(t, 2e-6), "GELU backward")


def test_patch_packing_is_the_4x4_stride_4_convolution():
    g = torch.Generator().manual_seed(1)
    B, S = 2, 32
    x = torch.randint(0, 256, (B, 3, S, S), generator=g, dtype=torch.uint8)
    w = torch.randn(96, 3, 4, 4, generator=g) * 0.05
    ref = F.conv2d(x.float(), w, None, 4)
    packed = torch.empty(B, S // 4, S // 4, 64, device=DEV)
    ops.pack_patches(x.to(DEV), packed)
    torch.cuda.synchronize()
    p = packed.cpu()
    assert p[..., 48:].abs().max() == 0
    got = torch.einsum("bhwk,ok->bohw", p[..., :48], w.view(96, 48))
    close(got, ref, 1e-5, "patch embedding as a 1x1 convolution over packed patches")


@pytest.mark.parametrize("dtype", DTYPES)
def test_patch_merging_gather_and_its_transpose(dtype):
    g = torch.Generator().manual_seed(2)
    B, H, Wd, C_ = 2, 6, 8, 32
    x = rq(torch.randn(B, H, Wd, C_, generator=g), dtype)
    ref = torch.cat([x[:, r::2, c::2, :] for c in range(2) for r in range(2)], dim=-1)       # SwinPatchMerging order
    fine = x.to(dtype).to(DEV).contiguous()
    merged = torch.empty(B, H // 2, Wd // 2, 4 * C_, dtype=dtype, device=DEV)
    ops.patch_merge(ops.view(fine), ops.view(merged))
    back = torch.empty_like(fine)
    ops.patch_merge(ops.view(back), ops.view(merged), reverse=True)
    torch.cuda.synchronize()
    assert torch.equal(merged.float().cpu(), ref) and torch.equal(back.float().cpu(), x)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,Wd,shift", [(14, 14, 0), (16, 16, 3), (9, 12, 3), (4, 4, 3)])
def test_window_partition_with_padding_and_shift(dtype, H, Wd, shift):
    """gather = pad to multiples of 7 (zeros) -> roll(-shift) -> window_partition; reverse = its inverse + residual add"""
    g = torch.Generator().manual_seed(H + shift)
    B, C_ = 2, 32
    x = rq(torch.randn(B, H, Wd, C_, generator=g), dtype)
    pr, pb = (7 - Wd % 7) % 7, (7 - H % 7) % 7
    h = F.pad(x, (0, 0, 0, pr, 0, pb))
    if shift:
        h = torch.roll(h, (-shift, -shift), (1, 2))
    ref = W.window_partition(h)                                   # [B*nW, 49, C]
    T = ref.shape[0] * 49
    xd = x.to(dtype).to(DEV).contiguous()
    win = torch.empty(1, T, 1, C_, dtype=dtype, device=DEV)
    ops.window_gather(ops.view(xd), ops.view(win), shift)
    res = rq(torch.randn(B, H, Wd, C_, generator=g), dtype)
    out = torch.empty_like(xd)
    ops.window_reverse(ops.view(win), ops.view(res.to(dtype).to(DEV).contiguous()), ops.view(out), shift)
    torch.cuda.synchronize()
    assert torch.equal(win.float().cpu().view(-1, 49, C_), ref)
    close(out.float().cpu(), rq(x + res, dtype) if dtype == torch.bfloat16 else x + res, 1e-6 if dtype == torch.float32 else 1e-2, "reverse + residual")


def _attention_reference(qkv, B, H, Wd, heads, shift, table):
    """the oracle's attention core on window tokens [nWin*49, 3C] (f64-capable)"""
    C_ = qkv.shape[1] // 3
    Hp, Wp = -(-H // 7) * 7, -(-Wd // 7) * 7
    q, k, v = (qkv[:, i * C_:(i + 1) * C_].reshape(-1, 49, heads, 32).transpose(1, 2) for i in range(3))
    bias = table[W.relative_position_index().view(-1)].view(49, 49, heads).permute(2, 0, 1).unsqueeze(0)
    att = (q @ k.transpose(-1, -2)) * 32 ** -0.5 + bias
    if shift:
        m = W.shift_mask(Hp, Wp, shift, att.dtype)
        att = (att.view(B, -1, heads, 49, 49) + m[None, :, None]).view(-1, heads, 49, 49)
    return (torch.softmax(att, -1) @ v).transpose(1, 2).reshape(-1, C_)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,Wd,heads,shift", [(14, 14, 3, 0), (16, 16, 3, 3), (9, 12, 6, 3), (7, 7, 12, 0), (140, 133, 3, 3)])
def test_window_attention_forward_and_backward(dtype, H, Wd, heads, shift):
    g = torch.Generator().manual_seed(H * heads + shift)
    B, C_ = 2, heads * 32
    nwin = B * (-(-H // 7)) * (-(-Wd // 7))
    T = nwin * 49
    qkv = rq(torch.randn(T, 3 * C_, generator=g), dtype).double().requires_grad_(True)
    table = (torch.randn(169, heads, generator=g) * 0.5).double().requires_grad_(True)
    ref = _attention_reference(qkv, B, H, Wd, heads, shift, table)
    go = rq(torch.randn(T, C_, generator=g), dtype).double()
    ref.backward(go)
    qv = tok(qkv.detach().float(), dtype)
    out = torch.empty(1, T, 1, C_, dtype=dtype, device=DEV)
    tb = table.detach().float().to(DEV).contiguous()
    ops.window_attention_fwd(qv, B, H, Wd, heads, shift, tb, ops.view(out))
    dqkv = torch.empty(1, T, 1, 3 * C_, dtype=dtype, device=DEV)
    dtable = torch.empty(169, heads, device=DEV)
    ws = torch.empty(ops.window_attention_bwd_workspace(B, H, Wd, heads), device=DEV)
    ops.window_attention_bwd(qv, tok(go.float(), dtype), B, H, Wd, heads, shift, tb, ops.view(dqkv), dtable, ws)
    torch.cuda.synchronize()
    t = 2e-5 if dtype == torch.float32 else 1e-2
    close(out.float().cpu().view(T, C_).double(), ref.detach(), t, "attention output")
    close(dqkv.float().cpu().view(T, 3 * C_).double(), qkv.grad, 5e-5 if dtype == torch.float32 else 2e-2, "dqkv")
    close(dtable.cpu().double(), table.grad, 1e-4 if dtype == torch.float32 else 2e-2, "relative-position table gradient")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,S", [(16, 1), (16, 2), (16, 3), (16, 6), (8, 3)])
def test_adaptive_average_pool(dtype, H, S):
    g = torch.Generator().manual_seed(H + S)
    B, C_ = 2, 64
    x = rq(torch.randn(B, C_, H, H, generator=g), dtype).requires_grad_(True)
    ref = F.adaptive_avg_pool2d(x, S)
    go = rq(torch.randn(ref.shape, generator=g), dtype)
    ref.backward(go)
    out = torch.empty(B, S, S, C_, dtype=dtype, device=DEV)
    ops.adaptive_avg_pool(ops.view(to_nhwc(x.detach(), dtype)), ops.view(out))
    gin = torch.empty(B, H, H, C_, dtype=dtype, device=DEV)
    ops.adaptive_avg_pool(ops.view(to_nhwc(go, dtype)), ops.view(gin), backward=True)
    torch.cuda.synchronize()
    t = 2e-6 if dtype == torch.float32 else 1e-2
    close(from_nhwc(out), ref.detach(), t, "adaptive pool")
    close(from_nhwc(gin), x.grad, t, "adaptive pool backward")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Hi,Ho", [(1, 16), (2, 16), (3, 16), (6, 16), (3, 8), (8, 16), (4, 32), (5, 7)])
def test_bilinear_resize_between_arbitrary_sizes(dtype, Hi, Ho):
    g = torch.Generator().manual_seed(Hi * 31 + Ho)
    B, C_ = 2, 32
    x = rq(torch.randn(B, C_, Hi, Hi + 1 if Hi > 1 else 1, generator=g), dtype).requires_grad_(True)
    Wi, Wo = x.shape[3], Ho + 2
    ref = F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False)
    go = rq(torch.randn(ref.shape, generator=g), dtype)
    ref.backward(go)
    base = rq(torch.randn(ref.shape, generator=g), dtype)
    out = to_nhwc(base, dtype)
    ops.resize_any(ops.view(to_nhwc(x.detach(), dtype)), ops.view(out), accumulate=True)
    plain = torch.empty(B, Ho, Wo, C_, dtype=dtype, device=DEV)
    ops.resize_any(ops.view(to_nhwc(x.detach(), dtype)), ops.view(plain))
    gin = torch.empty(B, Hi, Wi, C_, dtype=dtype, device=DEV)
    ops.resize_any(ops.view(to_nhwc(go, dtype)), ops.view(gin), backward=True)
    torch.cuda.synchronize()
    t = 2e-6 if dtype == torch.float32 else 1.5e-2
    close(from_nhwc(plain), ref.detach(), t, "resize")
    close(from_nhwc(out), ref.detach() + base, t, "resize + accumulate (FPN top-down add)")
    close(from_nhwc(gin), x.grad, t, "resize backward")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,M", [(96, 288, 150), (384, 96, 77), (96, 96, 260), (64, 96, 100)])
def test_linear_layers_with_widths_of_swin_stage_1(dtype, cin, cout, M):
    """nn.Linear as a 1x1 convolution with an output width that is no multiple of 64 (96 / 288): forward + bias, data gradient,
    weight gradient on the generic kernels with a masked last tile"""
    g = torch.Generator().manual_seed(cin + cout)
    x = rq(torch.randn(M, cin, generator=g), dtype).requires_grad_(True)
    w = rq(torch.randn(cout, cin, generator=g) / cin ** 0.5, dtype).requires_grad_(True)
    b = torch.randn(cout, generator=g)
    ref = F.linear(x, w, b)
    go = rq(torch.randn(M, cout, generator=g), dtype)
    ref.backward(go)
    xv, gv = tok(x.detach(), dtype), tok(go, dtype)
    wf, wd = ops.pack_conv_weight(w.detach().view(cout, cin, 1, 1).to(DEV).contiguous(), cin, dtype)
    y = torch.full((1, M, 1, cout), 7.0, dtype=dtype, device=DEV)
    ops.conv2d(xv, wf, b.to(DEV), ops.view(y), 1, 1)
    dx = torch.empty(1, M, 1, cin, dtype=dtype, device=DEV)
    ops.conv2d(gv, wd, None, ops.view(dx), 1, 1)
    dw = torch.empty(cout, cin, 1, 1, device=DEV)
    ws = torch.empty(ops.wgrad_workspace_floats_for(xv, gv, 1, 1, 1, 0), device=DEV)
    ops.conv2d_wgrad(xv, gv, dw, 1, 1, 1, 0, ws)
    torch.cuda.synchronize()
    t = 1e-4 if dtype == torch.float32 else 2e-2
    close(y.float().cpu().view(M, cout), ref.detach(), t, "linear forward")
    close(dx.float().cpu().view(M, cin), x.grad, t, "linear data gradient")
    close(dw.cpu().view(cout, cin), w.grad, 1e-4, "linear weight gradient")


# ---------------------------------------------------------------------------------------------------- the whole network
from cvcs_amd import nets, utils  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402


def rel_l2(got, ref):
    return ((got - ref).norm() / ref.norm().clamp_min(1e-12)).item()


def _build(NC, precision, seed=3):
    net = nets.SwinTUperNet(NC, precision)
    missing, unexpected = net.load_state_dict(W.init_params(NC, seed=seed), strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    return net.to(DEV)


def _f64_gradients(p32, img, lab, relu_masks):
    """float64 gradients of the oracle at the given parameters, with the HIP path's own ReLU decisions (the UPerNet units;
    the encoder has no decision anywhere: LayerNorm, softmax, GELU) - see tests/test_resnet_gpu.py"""
    p = {k: v.detach().double().clone() for k, v in p32.items()}
    names = [k for k in p if not W.is_buffer(k)]
    for k in names:
        p[k].requires_grad_(True)
    flips = []
    orig = F.relu
    it = iter(relu_masks)

    def forced(x, *a, **k):
        m = next(it)
        assert m.shape == x.shape, (m.shape, x.shape)
        bad = m != (x > 0)
        n = int(bad.sum())
        flips.append((n, (x.detach().abs()[bad].max() / x.detach().abs().max()).item() if n else 0.0))
        return x * m
    F.relu = forced
    try:
        loss = O.cross_entropy(W.forward(p, img.double(), train=True), lab.long(), None, 0)
    finally:
        F.relu = orig
    return dict(zip(names, torch.autograd.grad(loss, [p[k] for k in names]))), flips


def test_factory_name_and_parameter_names():
    net = utils.load_network({"net": "TSwin", "num_classes": 4, "precision": "bf16"}, DEV)   # configs/train/README.txt:44
    assert isinstance(net, nets.SwinTUperNet) and net.returns_logits and not net.requires_context
    assert sorted((k, tuple(v.shape)) for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")) == \
        sorted((k, tuple(s)) for k, s in W.param_spec(5))


@pytest.mark.parametrize("B,S,NC", [(4, 64, 5), (4, 96, 16)])
def test_fp32_train_steps_match_oracle(B, S, NC):
    """three SGD2 steps (S/train.py:121-126) on tiles whose token maps need window padding at every stage (16/8/4/2 and
    24/12/6/3 tokens a side against windows of 7): logits 1e-3 / loss 1e-4 against the f32 oracle; gradients 2e-4 relative L2
    per tensor against the float64 oracle evaluated at the HIP path's own ReLU decisions and the HIP network's own current
    parameters; the first update against the oracle trainer.  Batch 4: the 1x1 pyramid-pooling branch normalises over B values."""
    net = _build(NC, "fp32")
    tr = W.OracleTrainer(NC, "SGD2", ignore_index=0, seed=3)
    crit = utils.CrossEntropyLoss(ignore_index=0)
    optim, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 20}, net)
    img, lab = O.synthetic_tiles(B, S, NC, seed=5, structured=True)
    net.train()
    for step in range(3):
        p_before = {k: v.detach().cpu().clone() for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}
        with torch.no_grad():
            want_logits = W.forward({k: v.clone() for k, v in p_before.items()}, img.float(), train=True)
            want_loss = O.cross_entropy(want_logits, lab.long(), None, 0).item()
        logits = net(img.to(DEV), None)
        loss = crit(logits, lab.to(DEV))
        optim.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        masks = [from_nhwc(v.torch()) > 0 for v in net._engine.relu_order]
        got_logits = logits.detach().cpu().clone()
        grads = {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}
        optim.step()
        close(got_logits, want_logits, 1e-3, f"step {step} logits")
        assert abs(loss.item() - want_loss) <= 1e-4 * max(1.0, abs(want_loss)), (step, loss.item(), want_loss)
        g64, flips = _f64_gradients(p_before, img, lab, masks)
        assert all(mx <= 1e-4 for _, mx in flips), [f for f in flips if f[0]]
        # the 1x1 pyramid-pooling branch: a BatchNorm over B values per channel, its conv's gradient is a cancellation residue
        # gradients that are ZERO in exact arithmetic (a key bias shifts every score of a softmax row alike; the bias of an output
        # LayerNorm that only feeds 1x1 conv + train-mode BatchNorm is removed by the batch mean): float64 leaves ~1e-20, f32 its
        # own rounding residue - those are bounded in absolute terms against the typical gradient norm instead
        norms = sorted(g64[k].norm().item() for k in g64)
        gmax, gmed = norms[-1], norms[len(norms) // 2]
        zero = [k for k in g64 if g64[k].norm().item() <= 1e-9 * gmax]
        assert all(k.endswith(("k_proj.bias", "stage1.bias", "stage2.bias", "stage3.bias")) for k in zero), zero
        zworst = max((grads[k].norm().item() / gmed, k) for k in zero)
        e = sorted((rel_l2(grads[k].double(), g64[k]) * (0.04 if k.startswith("decode_head.psp_modules.0.") else 1.0), k)
                   for k in g64 if k not in zero)
        print(f"step {step}: {sum(n for n, _ in flips)} ReLU decisions differ from the float64 oracle's own; gradient rel-L2 vs "
              f"float64 at the HIP decisions: worst {e[-1]}, median {e[len(e) // 2][0]:.2e}; {len(zero)} exactly-zero gradients, "
              f"largest HIP residue {zworst[0]:.2e} of the median gradient norm ({zworst[1]})")
        assert e[-1][0] <= 2e-4, e[-5:]
        assert zworst[0] <= 1e-4, zworst
        if step == 0:
            tr.step(img, lab)
            sd = net.state_dict()
            worstp = max((rel_l2(sd[k].cpu(), tr.p[k].detach()), k) for k in tr.p)
            assert worstp[0] <= 1e-4, worstp
    net.eval()
    final = {k: v.detach().cpu().clone() for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}
    with torch.no_grad():
        ev = net(img.to(DEV), None).cpu()
        want = W.forward(final, img.float(), train=False)
        labels = net.predict_labels(img.to(DEV)).cpu()
    close(ev, want, 1e-3, "eval logits")
    assert torch.equal(labels.long(), ev.argmax(1))


def test_swin_base_variant_matches_oracle():
    """`BSwin` (configs/train/README.txt:45): Swin-B encoder (embed 128, depths 2/2/18/2, heads 4/8/16/32) + UPerNet - train-mode logits,
    loss and the first SGD2 update against the oracle in f32; eval logits; bf16 logits RMS"""
    NC, B, S = 5, 4, 64
    net = utils.load_network({"net": "BSwin", "num_classes": NC - 1, "precision": "fp32"}, DEV)
    assert isinstance(net, nets.SwinBUperNet) and utils.count_params(net) == 119_986_621 - 0   # Swin-B 88 M + UPerNet head
    p0 = W.init_params(NC, seed=3, variant="base")
    net.load_state_dict(p0, strict=False)
    tr = W.OracleTrainer(NC, "SGD2", ignore_index=0, seed=3, variant="base")
    crit = utils.CrossEntropyLoss(ignore_index=0)
    optim, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 20}, net)
    img, lab = O.synthetic_tiles(B, S, NC, seed=5, structured=True)
    net.train()
    logits = net(img.to(DEV), None)
    loss = crit(logits, lab.to(DEV))
    optim.zero_grad()
    loss.backward()
    optim.step()
    torch.cuda.synchronize()
    want_loss, want_logits, _ = tr.step(img, lab)
    close(logits.detach().cpu(), want_logits, 1e-3, "Swin-B train logits")
    assert abs(loss.item() - want_loss) <= 1e-4 * max(1.0, abs(want_loss))
    sd = net.state_dict()
    worst = max((rel_l2(sd[k].cpu(), tr.p[k].detach()), k) for k in tr.p)
    assert worst[0] <= 2e-4, worst
    net.eval()
    final = {k: v.detach().cpu().clone() for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}
    with torch.no_grad():
        ev = net(img.to(DEV), None).cpu()
        want = W.forward(final, img.float(), train=False)
    close(ev, want, 1e-3, "Swin-B eval logits")
    nb = nets.SwinBUperNet(NC, "bf16")
    nb.load_state_dict(p0, strict=False)
    nb = nb.to(DEV).eval()
    with torch.no_grad():
        evb = nb(img.to(DEV), None).cpu()
        want0 = W.forward(p0, img.float(), train=False)
    assert (evb - want0).pow(2).mean().sqrt().item() <= 1e-2 * want0.abs().max().item()


def test_fp32_forward_without_window_padding():
    """224 x 224: 56 / 28 / 14 / 7 tokens a side, the published geometry (no padding, the last stage a single window)"""
    net = _build(7, "fp32")
    img, _ = O.synthetic_tiles(2, 224, 7, seed=6, structured=True)
    p = W.init_params(7, seed=3)
    net.eval()
    with torch.no_grad():
        got = net(img.to(DEV), None).cpu()
        want = W.forward(p, img.float(), train=False)
    close(got, want, 1e-3, "eval logits 224")


def test_bf16_against_the_f32_path():
    """the benchmarked precision, 4 x 256 x 256 (bf16 storage of every activation, f32 accumulation, f32 LayerNorm / softmax / GELU
    arithmetic): logits against the f32 oracle; parameter gradients against the f32 HIP path's at identical parameters.
    The four pyramid-pooling branches are gated off for the gradient comparison (their BatchNorm gamma = 0 in both networks): they
    normalise B * s * s = 4 .. 144 values per channel whatever the tile size, and at B = 4 the 1x1 branch alone (4 nearly equal
    values, 1/sqrt(var + eps) up to 316) turns bf16 storage rounding into ~100 % noise on every encoder gradient
    (scripts/swin_precision_probe.py prints the per-tensor table with and without the gate).  What remains is the noise of the
    3x3 bottleneck's BatchNorm over 4 * 8 * 8 values and of the ReLU decisions that flip: ~10 % in the encoder, 3 % at the head."""
    NC, B, S = 16, 4, 256
    img, lab = O.synthetic_tiles(B, S, NC, seed=9, structured=True)
    out = {}
    for prec in ("fp32", "bf16"):
        net = nets.SwinTUperNet(NC, prec)
        sd = W.init_params(NC, seed=3)
        for i in range(4):
            sd[f"decode_head.psp_modules.{i}.1.batch_norm.weight"].zero_()
        net.load_state_dict(sd, strict=False)
        net = net.to(DEV)
        crit = utils.CrossEntropyLoss(ignore_index=0)
        net.train()
        logits = net(img.to(DEV), None)
        loss = crit(logits, lab.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        out[prec] = (logits.detach().cpu().clone(), {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}, loss.item())
    with torch.no_grad():
        want32 = W.forward(sd, img.float(), train=True)
    scale = want32.abs().max().item()
    close(out["fp32"][0], want32, 1e-3, "f32 logits")
    got = out["bf16"][0]
    emax, rms = (got - want32).abs().max().item() / scale, (got - want32).pow(2).mean().sqrt().item() / scale
    g32 = out["fp32"][1]
    gmax = max(v.norm().item() for v in g32.values())
    live = [k for k in g32 if g32[k].norm().item() > 1e-6 * gmax]
    e = sorted((rel_l2(out["bf16"][1][k], g32[k]), k) for k in live)
    print(f"bf16 logits / max|logit|: max {emax:.3e}, rms {rms:.3e}; loss {out['bf16'][2]:.5f} vs {out['fp32'][2]:.5f}")
    print("bf16 gradient rel-L2 vs the f32 path: worst", e[-3:], "median", e[len(e) // 2])
    assert rms <= 1e-2 and abs(out["bf16"][2] - out["fp32"][2]) <= 2e-2 * out["fp32"][2]
    assert e[len(e) // 2][0] <= 0.2 and e[-1][0] <= 0.6
    assert rel_l2(out["bf16"][1]["decode_head.classifier.weight"], g32["decode_head.classifier.weight"]) <= 5e-2


def test_bf16_end_to_end_inside_the_rounding_noise_floor():
    """train-mode logits of the bf16 path, 4 x 128 x 128, every branch on: the yardstick is the network's own rounding-noise floor - the
    distance between the f32 oracle and its bf16-EMULATION mode (rounding where the HIP path stores bf16, f32 arithmetic), measured on the
    CPU.  The HIP path must stay within 1.5x of it (max and RMS) and must be no further from the emulation than the emulation is from f32."""
    NC, B, S = 16, 4, 128
    img, _ = O.synthetic_tiles(B, S, NC, seed=9, structured=True)
    net = _build(NC, "bf16")
    net.train()
    with torch.no_grad():
        got = net(img.to(DEV), None).cpu()
        p = W.init_params(NC, seed=3)
        want32 = W.forward(p, img.float(), train=True)
        emu = W.forward(p, img.float(), train=True, emulate_bf16=True)
    scale = want32.abs().max().item()
    floor, floor_rms = (emu - want32).abs().max().item() / scale, (emu - want32).pow(2).mean().sqrt().item() / scale
    e32, rms32 = (got - want32).abs().max().item() / scale, (got - want32).pow(2).mean().sqrt().item() / scale
    rms_emu = (got - emu).pow(2).mean().sqrt().item() / scale
    print(f"bf16 logits / max|logit|: max {e32:.3e}, rms {rms32:.3e} from the f32 oracle (emulation: max {floor:.3e}, rms {floor_rms:.3e}); "
          f"rms {rms_emu:.3e} from the emulation")
    assert e32 <= 1.5 * floor and rms32 <= 1.5 * floor_rms and rms_emu <= 1.5 * floor_rms


def test_swin_trains_reproducibly():
    img, lab = O.synthetic_tiles(4, 128, 5, seed=2, structured=True)
    runs = []
    for _ in range(2):
        net = _build(5, "bf16")
        crit = utils.CrossEntropyLoss(ignore_index=0)
        optim, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 20}, net)
        net.train()
        losses = []
        for _ in range(10):
            loss = crit(net(img.to(DEV), None), lab.to(DEV))
            optim.zero_grad()
            loss.backward()
            optim.step()
            losses.append(loss.item())
        runs.append((losses, net.flat_parameters()[0].clone()))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    assert runs[0][0][-1] < 0.8 * runs[0][0][0], runs[0][0]
