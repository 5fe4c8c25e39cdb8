"""GPU parity of the reference's `MobileNet` factory name (S/nets.py:277-311, S/utils.py:183-184: torchvision's deeplabv3_mobilenet_v3_large):
the new kernels (csrc/mobilenet.hip, the hardswish codes of the BatchNorm passes) against ATen CPU ops, the network against
`oracle/mobilenet_oracle.py` (PARITY UNPINNED: torchvision is absent from this image; see that file)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cvcs_amd import nets, ops, utils  # noqa: E402
from oracle import mobilenet_oracle as M  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402

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


def rel_l2(got, ref):
    return ((got - ref).norm() / ref.norm().clamp_min(1e-12)).item()


# ---------------------------------------------------------------------------------------------------- kernels
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C_,K,stride,dil,H,W", [(64, 3, 1, 1, 12, 9), (96, 3, 2, 1, 16, 16), (96, 5, 2, 1, 16, 12), (128, 5, 1, 1, 7, 9), (224, 3, 1, 1, 6, 6),
                                                  (672, 5, 1, 2, 8, 8), (960, 5, 1, 2, 4, 4), (32, 3, 1, 1, 33, 17)])
def test_depthwise_forward_statistics_data_and_weight_gradient(dtype, C_, K, stride, dil, H, W):
    """nn.Conv2d(C, C, K, stride, (K-1)//2*dil, dil, groups=C): forward + the BatchNorm statistics of its epilogue, data gradient, weight gradient"""
    g = torch.Generator().manual_seed(C_ + K)
    B, pad = 3, (K - 1) // 2 * dil
    x = rq(torch.randn(B, C_, H, W, generator=g), dtype).requires_grad_(True)
    w = (torch.randn(C_, 1, K, K, generator=g) / K).requires_grad_(True)
    ref = F.conv2d(x, w, None, stride, pad, dil, groups=C_)
    go = rq(torch.randn(ref.shape, generator=g), dtype)
    ref.backward(go)
    Ho, Wo = ref.shape[2:]
    wt = w.detach().view(C_, K * K).t().contiguous().to(DEV)
    xd, y = to_nhwc(x.detach(), dtype), torch.empty(B, Ho, Wo, C_, dtype=dtype, device=DEV)
    rows = ops.dwconv_rows(B * Ho * Wo, C_, dtype)
    stats = (torch.zeros(rows * C_, device=DEV), torch.zeros(rows * C_, device=DEV), torch.zeros(rows, device=DEV))
    ops.dwconv(ops.view(xd), wt, K, stride, pad, dil, ops.view(y), stats)
    scale, shift, mean, invstd = (torch.empty(C_, device=DEV) for _ in range(4))
    rm, rv = torch.zeros(C_, device=DEV), torch.ones(C_, device=DEV)
    ops.bn_finalize(stats, rows, B * Ho * Wo, C_, torch.ones(C_, device=DEV), torch.zeros(C_, device=DEV), rm, rv, True, scale, shift, mean, invstd,
                    momentum=0.01, eps=1e-3)
    dx = torch.empty_like(xd)
    god = to_nhwc(go, dtype)
    ops.dwconv_dgrad(ops.view(god), wt, K, stride, pad, dil, ops.view(dx))
    prow = ops.dwconv_wgrad_rows(B * Ho * Wo, C_, K, dtype)
    part = torch.zeros(prow * K * K * C_, device=DEV)
    ops.dwconv_wgrad(ops.view(xd), ops.view(god), K, stride, pad, dil, part)
    gw = torch.empty(K * K, C_, device=DEV)
    ops.colsum_finalize(part, prow, K * K * C_, gw)
    torch.cuda.synchronize()
    t = 1e-5 if dtype == torch.float32 else 1e-2
    close(from_nhwc(y), ref.detach(), t, "forward")
    stored = from_nhwc(y)
    close(mean.cpu(), stored.mean(dim=(0, 2, 3)), 1e-4, "batch mean of the stored values")
    close(invstd.cpu(), 1.0 / torch.sqrt(stored.var(dim=(0, 2, 3), unbiased=False) + 1e-3), 1e-4, "batch invstd")
    close(rv.cpu(), 0.99 + 0.01 * stored.var(dim=(0, 2, 3), unbiased=True), 1e-4, "running variance (momentum 0.01)")
    close(from_nhwc(dx), x.grad, t, "data gradient")
    close(gw.cpu().t().reshape(C_, 1, K, K), w.grad, 1e-4 if dtype == torch.float32 else 1e-4, "weight gradient")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C_", [32, 96, 160, 224, 960])
def test_batchnorm_hardswish_forward_and_backward_any_width(dtype, C_):
    """BatchNorm (train) -> nn.Hardswish on widths that are neither powers of two nor multiples of 256 (cvcs_bn_act act 2, cvcs_bn_bwd_* mode 3)"""
    g = torch.Generator().manual_seed(C_)
    B, H, W = 3, 7, 5
    y = rq(torch.randn(B, C_, H, W, generator=g) * 2.0 + 0.5, dtype).requires_grad_(True)
    gamma = (0.5 + torch.rand(C_, generator=g)).requires_grad_(True)
    beta = ((torch.rand(C_, generator=g) - 0.5)).requires_grad_(True)
    out = F.hardswish(F.batch_norm(y, None, None, gamma, beta, True, 0.01, 1e-3))
    go = rq(torch.randn(out.shape, generator=g), dtype)
    out.backward(go)
    M_ = B * H * W
    yd, gd = to_nhwc(y.detach(), dtype), to_nhwc(go, dtype)
    mean = y.detach().mean(dim=(0, 2, 3))
    invstd = 1.0 / torch.sqrt(y.detach().var(dim=(0, 2, 3), unbiased=False) + 1e-3)
    scale = (gamma.detach() * invstd).to(DEV)
    shift = (beta.detach() - mean * gamma.detach() * invstd).to(DEV)
    a = torch.empty_like(yd)
    ops.bn_act(ops.view(yd), scale, shift, ops.HARDSWISH, ops.view(a))
    rows = ops.bn_bwd_rows(M_)
    p0, p1, p2 = (torch.zeros(rows * C_, device=DEV) for _ in range(3))
    md, isd = mean.to(DEV), invstd.to(DEV)
    ops.bn_bwd_reduce(ops.view(yd), ops.view(gd), None, scale, shift, md, isd, ops.BN_MODE_HARDSWISH, p0, p1)
    dgamma, dbeta, ca, cb = (torch.empty(C_, device=DEV) for _ in range(4))
    ops.bn_bwd_finalize(p0, p1, rows, M_, C_, gamma.detach().to(DEV), isd, dgamma, dbeta, ca, cb)
    dy = torch.empty_like(yd)
    ops.bn_bwd_apply(ops.view(yd), ops.view(gd), None, scale, shift, md, isd, ca, cb, ops.BN_MODE_HARDSWISH, ops.view(dy), p2)
    torch.cuda.synchronize()
    t = 1e-5 if dtype == torch.float32 else 1e-2
    close(from_nhwc(a), out.detach(), t, "hardswish(bn(y))")
    close(dgamma.cpu(), gamma.grad, 1e-4, "dgamma")
    close(dbeta.cpu(), beta.grad, 1e-4, "dbeta")
    close(from_nhwc(dy), y.grad, 1e-4 if dtype == torch.float32 else 1e-2, "dy")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C_,H,W", [(96, 8, 8), (128, 5, 7), (480, 4, 4), (960, 2, 3)])
def test_squeeze_excite_passes(dtype, C_, H, W):
    """x * s[b, c] forward; backward: ds = sum_p g * x, dx = g * s + gpool / HW; hardsigmoid forward / backward; image mean on odd widths"""
    g = torch.Generator().manual_seed(C_)
    B = 3
    x = rq(torch.randn(B, C_, H, W, generator=g), dtype)
    t = rq(torch.randn(B, C_, 1, 1, generator=g) * 3, dtype).requires_grad_(True)
    s = F.hardsigmoid(t)
    go = rq(torch.randn(B, C_, H, W, generator=g), dtype)
    gp = rq(torch.randn(B, C_, 1, 1, generator=g), dtype)
    xd, td, gd, gpd = to_nhwc(x, dtype), to_nhwc(t.detach(), dtype), to_nhwc(go, dtype), to_nhwc(gp, dtype)
    sd = torch.empty_like(td)
    ops.hardsigmoid(ops.view(td), ops.view(sd))
    out = torch.empty_like(xd)
    ops.se_scale(ops.view(xd), ops.view(sd), ops.view(out))
    ds = torch.empty_like(td)
    ops.image_dot(ops.view(gd), ops.view(xd), ops.view(ds))
    gt = torch.empty_like(td)
    ops.hardsigmoid(ops.view(td), ops.view(gt), g=ops.view(ds))
    dx = torch.empty_like(xd)
    ops.se_scale(ops.view(gd), ops.view(sd), ops.view(dx), add=ops.view(gpd), add_scale=1.0 / (H * W))
    pooled = torch.empty_like(td)
    ops.image_sum(ops.view(xd), ops.view(pooled), 1.0 / (H * W))
    torch.cuda.synchronize()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    sq = rq(s.detach(), dtype)
    close(from_nhwc(sd), s.detach(), tol, "hardsigmoid")
    close(from_nhwc(out), x * sq, tol, "scale")
    want_ds = (go * x).sum(dim=(2, 3), keepdim=True)
    close(from_nhwc(ds), want_ds, tol, "gradient of the scale")
    (s * rq(want_ds, dtype)).sum().backward()
    close(from_nhwc(gt), t.grad, tol, "hardsigmoid backward")
    close(from_nhwc(dx), go * sq + gp / (H * W), tol, "gradient of x")
    close(from_nhwc(pooled), x.mean(dim=(2, 3), keepdim=True), tol, "image mean")


@pytest.mark.parametrize("dtype", DTYPES)
def test_batchnorm_apply_with_residual_add(dtype):
    g = torch.Generator().manual_seed(1)
    B, C_, H, W = 2, 96, 5, 6
    y, r = rq(torch.randn(B, C_, H, W, generator=g), dtype), rq(torch.randn(B, C_, H, W, generator=g), dtype)
    sc, sh = torch.randn(C_, generator=g), torch.randn(C_, generator=g)
    out = torch.empty(B, H, W, C_, dtype=dtype, device=DEV)
    ops.bn_add(ops.view(to_nhwc(y, dtype)), sc.to(DEV), sh.to(DEV), ops.view(to_nhwc(r, dtype)), ops.view(out))
    out2 = torch.empty_like(out)
    ops.bn_add(ops.view(to_nhwc(y, dtype)), sc.to(DEV), sh.to(DEV), None, ops.view(out2))
    torch.cuda.synchronize()
    t = 1e-6 if dtype == torch.float32 else 1e-2
    close(from_nhwc(out), y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + r, t, "bn + residual")
    close(from_nhwc(out2), y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), t, "bn alone")


# ---------------------------------------------------------------------------------------------------- the network
def _build(NC, precision, seed=3, aspp_dropout=0.0):
    net = nets.DeepLabV3MobileNet(NC, precision, aspp_dropout=aspp_dropout)
    sd = {net._to_reference(k): v for k, v in M.init_params(NC, seed=seed).items()}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    return net.to(DEV)


def _oracle_named(net):
    return {net._from_reference(k): v.detach().cpu().clone() for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}


def test_factory_name_and_reference_checkpoint_keys(tmp_path):
    """S/utils.py:183-184; a checkpoint as the reference writes it (torchvision key names under `model.`, `module.` prefix, the unused
    auxiliary head) loads through `wrapper` / `custom_load` (S/nets.py:306-311)"""
    net = utils.load_network({"net": "MobileNet", "num_classes": 15, "precision": "bf16"}, DEV)
    assert isinstance(net, nets.DeepLabV3MobileNet) and net.wrapper and net.returns_logits and not net.requires_context
    want = sum(int(torch.tensor(s).prod()) for k, s in M.param_spec(16) if not M.is_buffer(k))
    assert utils.count_params(net) == want
    sd = net.state_dict()
    for k, shape in (("model.backbone.0.0.weight", (16, 3, 3, 3)), ("model.backbone.1.block.0.0.weight", (16, 1, 3, 3)),
                     ("model.backbone.4.block.2.fc1.weight", (24, 72, 1, 1)), ("model.backbone.4.block.2.fc2.bias", (72,)),
                     ("model.backbone.13.block.1.0.weight", (672, 1, 5, 5)), ("model.backbone.16.0.weight", (960, 160, 1, 1)),
                     ("model.backbone.16.1.running_var", (960,)), ("model.classifier.0.convs.3.0.weight", (256, 960, 3, 3)),
                     ("model.classifier.0.project.0.weight", (256, 1280, 1, 1)), ("model.classifier.1.weight", (256, 256, 3, 3)),
                     ("model.classifier.4.weight", (16, 256, 1, 1))):
        assert k in sd and tuple(sd[k].shape) == shape, (k, shape)
    ck = {"model_state_dict": {"module." + k: (v.float() * 0 + 0.25 if v.dtype == torch.float32 else v).cpu() for k, v in sd.items()}}
    ck["model_state_dict"]["module.model.aux_classifier.4.weight"] = torch.zeros(21, 10, 1, 1)
    path = str(tmp_path / "checkpoint1")
    torch.save(ck, path)
    other = utils.load_network({"net": "MobileNet", "num_classes": 15, "precision": "bf16"}, DEV)
    utils.load_checkpoint({"load_checkpoint": path}, other)
    out = other.state_dict()
    assert all((v == 0.25).all() for k, v in out.items() if v.dtype == torch.float32 and "aux_classifier" not in k)
    assert (out["model.aux_classifier.4.weight"] == 0).all()


def _f64_gradients(p32, img, lab, relu_masks, hs_acts):
    """float64 gradients of the oracle with every F.relu decision forced to the given masks (execution order; the HIP maps carry padding
    channels beyond the oracle's) and every F.hardswish evaluated on the branch (0 below -3 | x (x + 3) / 6 | x above 3) the HIP path's stored
    output says it took (exactly 0 / above 3 / neither: the function is continuous, its derivative jumps by 1/2 at both knots); also (count,
    largest relative distance from the kink) of the decisions that differ from the oracle's own"""
    p = {k: v.detach().double().clone() for k, v in p32.items()}
    names = [k for k in p if not M.is_buffer(k)]
    for k in names:
        p[k].requires_grad_(True)
    flips, orig, it = [], F.relu, iter(relu_masks)

    def forced(x, *a, **k):
        m = next(it)[:, :x.shape[1]]
        assert m.shape == x.shape, (m.shape, x.shape)
        bad = m != (x > 0)
        n = int(bad.sum())
        flips.append((n, (x.detach().abs()[bad].max() / x.detach().abs().max()).item() if n else 0.0))
        return x * m
    orig_hs, ih = F.hardswish, iter(hs_acts)

    def forced_hs(x, *a, **k):
        act = next(ih)[:, :x.shape[1]]
        assert act.shape == x.shape, (act.shape, x.shape)
        low, high = act == 0, act > 3
        own_low, own_high = x < -3, x > 3
        bad = (low != own_low) | (high != own_high)
        n = int(bad.sum())
        flips.append((n, (torch.minimum((x.detach() + 3).abs(), (x.detach() - 3).abs())[bad].max() / x.detach().abs().max()).item() if n else 0.0))
        return torch.where(low, x * 0, torch.where(high, x, x * (x + 3) / 6))
    F.relu, F.hardswish = forced, forced_hs
    try:
        loss = O.cross_entropy(M.forward(p, img.double(), train=True), lab.long(), None, 0)
    finally:
        F.relu, F.hardswish = orig, orig_hs
    assert next(it, None) is None and next(ih, None) is None, "the oracle took fewer activations than the HIP path"
    return dict(zip(names, torch.autograd.grad(loss, [p[k] for k in names]))), flips


@pytest.mark.parametrize("B,S,NC", [(4, 64, 5), (3, 128, 16)])
def test_fp32_train_steps_match_oracle(B, S, NC):
    """three SGD2 steps (S/train.py:121-126) at the HIP network's own current parameters: logits / loss against the f32 oracle, every
    parameter gradient against the oracle's autograd (relative L2 per tensor), the running statistics (momentum 0.01, eps 1e-3 in the
    backbone), the first update against the oracle trainer; then eval-mode logits and the fused arg-max"""
    net = _build(NC, "fp32")
    tr = M.OracleTrainer(NC, "SGD2", ignore_index=0, seed=3)
    crit = utils.CrossEntropyLoss(ignore_index=0)
    optim, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 20}, net)
    img, lab = O.synthetic_tiles(B, S, NC, seed=5, structured=True)
    net.train()
    for step in range(3):
        _before = _oracle_named(net)
        p = {k: v.clone() for k, v in _before.items()}
        names = [k for k in p if not M.is_buffer(k)]
        for k in names:
            p[k].requires_grad_(True)
        want_logits = M.forward(p, img.float(), train=True)
        want_loss = O.cross_entropy(want_logits, lab.long(), None, 0)
        want_grads = dict(zip(names, torch.autograd.grad(want_loss, [p[k] for k in names])))
        logits = net(img.to(DEV), None)
        loss = crit(logits, lab.to(DEV))
        optim.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        masks = [from_nhwc(v.torch()) > 0 for v in net._engine.relu_order]
        hs_acts = [from_nhwc(v.torch()) for v in net._engine.hs_order]
        got_logits = logits.detach().cpu().clone()
        grads = {net._from_reference(k): q.grad.detach().cpu().clone() for k, q in net.named_parameters()}
        after = _oracle_named(net)
        close(got_logits, want_logits.detach(), 1e-3, f"step {step} logits")
        assert abs(loss.item() - want_loss.item()) <= 1e-4 * max(1.0, abs(want_loss.item())), (step, loss.item(), want_loss.item())
        for k in p:
            if M.is_buffer(k):      # the oracle's forward has updated its copy in place
                assert rel_l2(after[k], p[k].detach()) <= 1e-4, (step, k)
        # every tensor against the float64 oracle evaluated at the HIP path's own ReLU decisions (see tests/test_resnet_gpu.py: on the 4x4 ... 8x8
        # maps of these tiles ONE differing decision moves a BatchNorm's bias gradient by a percent, and everything upstream with it); the
        # differing decisions are counted and each must sit on a pre-activation below 1e-4 of its tensor's max.  The projection BatchNorms'
        # bias gradients vanish analytically (the BatchNorm behind the next 1x1 convolution removes a per-channel shift): skipped by norm.
        g64, flips = _f64_gradients(_before, img, lab, masks, hs_acts)
        assert all(mx <= 1e-4 for _, mx in flips), [f for f in flips if f[0]]
        rows = sorted((rel_l2(grads[k].double(), g64[k]) * (0.04 if k.startswith("aspp.convs.4.") else 1.0), k) for k in names if g64[k].norm() > 1e-9)
        print(f"step {step}: {sum(n for n, _ in flips)} activation decisions differ from the float64 oracle's own; gradient rel-L2 vs float64 at the HIP "
              f"decisions (ReLU masks, hardswish branches): worst {rows[-1]}, median {rows[len(rows) // 2][0]:.2e}")
        assert rows[-1][0] <= 2e-4, rows[-3:]
        optim.step()
        if step == 0:
            tr.step(img, lab)
            sd = _oracle_named(net)
            worstp = max((rel_l2(sd[k], tr.p[k].detach()), k) for k in tr.p)
            assert worstp[0] <= 5e-4, worstp
    net.eval()
    final = _oracle_named(net)
    with torch.no_grad():
        ev = net(img.to(DEV), None).cpu()
        want = M.forward({k: v.clone() for k, v in final.items()}, img.float(), train=False)
        want64 = M.forward({k: v.double() for k, v in final.items()}, img.double(), train=False)
        labels = net.predict_labels(img.to(DEV)).cpu()
    # three steps at momentum 0.01 leave the running statistics near their initial (0, 1): the eval-mode network barely normalises the raw
    # 0..255 input, activations reach 1e4 and torch-CPU f32 itself sits up to 3e-3 from float64 - the HIP path is held to 1e-3 or 3x that
    scale = want64.abs().max().item()
    e_hip, e_ref = (ev.double() - want64).abs().max().item() / scale, (want.double() - want64).abs().max().item() / scale
    print(f"eval logits vs float64 (fraction of max|logit|): HIP {e_hip:.2e}, torch-CPU f32 {e_ref:.2e}")
    assert e_hip <= max(1e-3, 3.0 * e_ref), (e_hip, e_ref)
    assert torch.equal(labels.long(), ev.argmax(1))


def test_padding_channels_stay_zero_and_master_gradients_are_complete():
    """the engine's padded mirrors: after a step, every padding lane of the mirrored BatchNorm vectors / gradients is exactly zero and no master
    gradient is left unwritten (NaN-poisoned before the step)"""
    NC = 5
    net = _build(NC, "fp32")
    img, lab = O.synthetic_tiles(4, 64, NC, seed=7, structured=True)
    crit = utils.CrossEntropyLoss(ignore_index=0)
    net.train()
    loss = crit(net(img.to(DEV), None), lab.to(DEV))
    net._flat_grad.fill_(float("nan"))
    loss.backward()
    torch.cuda.synchronize()
    bad = [k for k, q in net.named_parameters() if not torch.isfinite(q.grad).all()]
    assert not bad, bad
    eng = net._engine
    for name, g in eng.G.items():
        if name in eng.MG and g.data_ptr() != eng.MG[name].data_ptr() and g.dim() == 1:
            n = eng.MG[name].numel()
            assert (g[n:] == 0).all(), name


def test_bf16_forward_within_the_rounding_noise_floor_and_trains():
    """the benchmarked precision: train-mode logits within 3x the distance between the f32 oracle and its own bf16 emulation; 20 SGD steps
    on one batch bring the loss down like the f32 network's"""
    NC, B, S = 8, 4, 128
    img, lab = O.synthetic_tiles(B, S, NC, seed=9, structured=True)
    p = M.init_params(NC, seed=3)
    with torch.no_grad():
        ref = M.forward({k: v.clone() for k, v in p.items()}, img.float(), train=True)
        emu = M.forward({k: v.clone() for k, v in p.items()}, img.float(), train=True, emulate_bf16=True)
    floor = rel_l2(emu, ref)
    curves = {}
    for precision in ("bf16", "fp32"):
        net = _build(NC, precision)
        net.train()
        first = net(img.to(DEV), None).detach().cpu()
        if precision == "bf16":
            err = rel_l2(first, ref)
            print(f"bf16 train-mode logits: rel-L2 {err:.3e} vs the f32 oracle (its own bf16 emulation: {floor:.3e})")
            assert err <= 3.0 * floor + 1e-3
        crit = utils.CrossEntropyLoss(ignore_index=0)
        optim, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 20}, net)
        losses = []
        for _ in range(20):
            loss = crit(net(img.to(DEV), None), lab.to(DEV))
            optim.zero_grad()
            loss.backward()
            optim.step()
            losses.append(loss.item())
        curves[precision] = losses
    print("loss curves", {k: [round(v[0], 4), round(v[-1], 4)] for k, v in curves.items()})
    assert curves["fp32"][-1] < 0.7 * curves["fp32"][0]
    assert abs(curves["bf16"][0] - curves["fp32"][0]) <= 2e-2 * curves["fp32"][0]
    assert curves["bf16"][-1] < 0.75 * curves["bf16"][0]


def test_aspp_dropout_mask_is_the_replayable_counter_hash():
    """train mode with the reference's Dropout(0.5) behind the ASPP projection: the logits equal the oracle's under the host-replayed mask"""
    NC, B, S = 5, 4, 64
    net = _build(NC, "fp32", aspp_dropout=0.5)
    img, _ = O.synthetic_tiles(B, S, NC, seed=5, structured=True)
    net.train()
    p = _oracle_named(net)
    logits = net(img.to(DEV), None).detach().cpu()
    h = S // 16
    keep = ops.dropout_mask_host(net.dropout_seed, 0, B * h * h * 256, 0.5).view(B, h, h, 256).permute(0, 3, 1, 2)
    with torch.no_grad():
        want = M.forward({k: v.clone() for k, v in p.items()}, img.float(), train=True, aspp_keep=keep, aspp_dropout=0.5)
        plain = M.forward({k: v.clone() for k, v in p.items()}, img.float(), train=True)
    close(logits, want, 1e-3, "logits under the replayed mask")
    assert rel_l2(logits, plain) > 0.05


def test_bf16_layer_by_layer_from_the_stored_operands():
    """the benchmarked precision, 4 x 256 x 256: every convolution output (GEMM 1x1 with padded widths, depthwise, atrous, the stem as patch
    GEMM) recomputed in f32 from the path's OWN stored bf16 input and the master weights (bf16-rounded where the kernel reads bf16 operands;
    the depthwise filters stay f32) must agree to bf16 storage rounding; the padding channels of every map are exactly zero"""
    NC, B, S = 16, 4, 256
    net = _build(NC, "bf16")
    img, _ = O.synthetic_tiles(B, S, NC, seed=9, structured=True)
    net.train()
    net(img.to(DEV), None)
    torch.cuda.synchronize()
    eng = net._engine
    worst, n_dw, n_pad = (0.0, ""), 0, 0
    for conv, (u, act_out) in eng.units.items():
        w = eng.MP[conv + ".weight"].detach().cpu()
        cout, cin_g, k, _ = w.shape
        dw = getattr(u, "dw", False)
        got = from_nhwc(u.y.torch())
        if conv == "encoder.0.0":       # the stem: its stored operand is the patch matrix of the raw tile
            ref = F.conv2d(img.float(), w.to(torch.bfloat16).float(), None, 2, 1)
        else:
            x = from_nhwc(u.x.torch())
            if dw:
                ref = F.conv2d(x[:, :cout], w, None, u.stride, u.pad, u.dil, groups=cout)
                n_dw += 1
            else:
                ref = F.conv2d(x[:, :cin_g], w.to(torch.bfloat16).float(), None, u.stride, u.pad, u.dil)
        if got.shape[1] > cout:
            assert (got[:, cout:] == 0).all(), f"{conv}: padding channels of the output are not zero"
            n_pad += 1
        e = (got[:, :cout] - ref).abs().max().item() / ref.abs().max().item()
        worst = max(worst, (e, conv))
        assert e <= 2.0 ** -8, f"{conv}: conv output off by {e:.3e} of its max"
    assert n_dw == 15 and n_pad >= 20 and len(eng.units) == 1 + 15 * 2 + 14 + 1 + 7
    print(f"MobileNet {B}x{S}: worst conv output error {worst[0]:.3e} ({worst[1]}) of its max [bf16 ulp = 3.9e-3]; {n_pad} padded maps with exactly-zero padding")


def test_mobilenet_trains_reproducibly():
    """fixed-order reductions everywhere (statistics, weight-gradient partials, the gather / scatter tables): two runs agree bit for bit"""
    img, lab = O.synthetic_tiles(4, 128, 5, seed=2, structured=True)
    runs = []
    for _ in range(2):
        net = _build(5, "bf16", aspp_dropout=0.5)
        crit = utils.CrossEntropyLoss(ignore_index=0)
        optim, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 20}, net)
        net.train()
        losses = []
        for _ in range(8):
            loss = crit(net(img.to(DEV), None), lab.to(DEV))
            optim.zero_grad()
            loss.backward()
            optim.step()
            losses.append(loss.item())
        runs.append((losses, net.flat_parameters()[0].clone()))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    assert runs[0][0][-1] < runs[0][0][0], runs[0][0]
