"""GPU parity of the DeepLab path (BASELINE.json configs[2] DeepLabV3+; the reference's `Resnet101` = DeepLabV3) - kernels
against the ATen CPU ops, the networks against `oracle/deeplab_oracle.py` (PARITY UNPINNED: torchvision, whose DeepLabV3 the
reference wraps, is absent from this image; see that file).  Tolerances as in tests/test_resnet_gpu.py."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cvcs_amd import nets, ops, utils  # noqa: E402
from cvcs_amd.ops import View  # noqa: E402
from oracle import deeplab_oracle as D  # noqa: E402
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
@pytest.mark.parametrize("scale,H,W", [(4, 8, 8), (2, 5, 7), (8, 4, 4), (3, 6, 5)])
def test_bilinear_resize_nhwc_forward_and_backward(dtype, scale, H, W):
    """F.interpolate(scale_factor, mode='bilinear', align_corners=False) and its autograd backward (as a gather)"""
    g = torch.Generator().manual_seed(scale)
    B, C_ = 2, 32
    x = rq(torch.randn(B, C_, H, W, generator=g), dtype).requires_grad_(True)
    ref = F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=False)
    go = rq(torch.randn(ref.shape, generator=g), dtype)
    ref.backward(go)
    out = torch.empty(B, H * scale, W * scale, C_, dtype=dtype, device=DEV)
    ops.resize_bilinear_fwd(ops.view(to_nhwc(x.detach(), dtype)), ops.view(out), scale)
    gin = torch.empty(B, H, W, C_, dtype=dtype, device=DEV)
    ops.resize_bilinear_bwd(ops.view(to_nhwc(go, dtype)), ops.view(gin), scale)
    torch.cuda.synchronize()
    t = 1e-6 if dtype == torch.float32 else 1e-2
    close(from_nhwc(out), ref.detach(), t, "resize forward")
    close(from_nhwc(gin), x.grad, t, "resize backward")


@pytest.mark.parametrize("scale", [4, 8, 2, 3, 16])
def test_bilinear_resize_of_the_logits(scale):
    g = torch.Generator().manual_seed(scale)
    z = torch.randn(2, 5, 9, 6, generator=g).requires_grad_(True)
    ref = F.interpolate(z, scale_factor=scale, mode="bilinear", align_corners=False)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    out = torch.empty(ref.shape, device=DEV)
    gin = torch.empty(z.shape, device=DEV)
    ops.resize_bilinear_nchw_fwd(z.detach().to(DEV), out, scale)
    ops.resize_bilinear_nchw_bwd(go.to(DEV), gin, scale)
    torch.cuda.synchronize()
    close(out.cpu(), ref.detach(), 1e-6, "logit resize forward")
    close(gin.cpu(), z.grad, 1e-5, "logit resize backward")


@pytest.mark.parametrize("dtype", DTYPES)
def test_image_pooling_mean_and_broadcast(dtype):
    g = torch.Generator().manual_seed(3)
    B, C_, H, W, off = 3, 256, 5, 7, 64
    x = rq(torch.randn(B, C_, H, W, generator=g), dtype)
    wide = torch.zeros(B, H, W, C_ + off, dtype=dtype, device=DEV)
    wide[..., off:] = to_nhwc(x, dtype)
    out = torch.empty(B, 1, 1, C_, dtype=dtype, device=DEV)
    ops.image_sum(View(wide, off, C_), ops.view(out), 1.0 / (H * W))
    bc = torch.zeros(B, H, W, C_ + off, dtype=dtype, device=DEV)
    ops.image_broadcast(ops.view(out), View(bc, off, C_), 0.5)
    torch.cuda.synchronize()
    close(from_nhwc(out), x.mean(dim=(2, 3), keepdim=True), 1e-6 if dtype == torch.float32 else 1e-2, "per-image mean")
    assert torch.equal(bc[..., :off].cpu(), torch.zeros(B, H, W, off, dtype=dtype))
    close(from_nhwc(bc[..., off:]), 0.5 * from_nhwc(out).expand(-1, -1, H, W), 1e-6 if dtype == torch.float32 else 1e-2, "broadcast")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C_,NC,H,W", [(256, 21, 9, 13), (64, 5, 9, 13), (512, 16, 9, 13),
                                       # H * W a multiple of 32 and C of 64: the bf16 backward runs on the matrix cores
                                       (512, 16, 8, 16), (256, 21, 16, 16), (64, 5, 4, 8), (1024, 3, 8, 8), (192, 32, 8, 12)])
def test_linear_head_any_channel_count(dtype, C_, NC, H, W):
    """the 1x1 classifier (S/nets.py:243-244) on C channels: logits, fused argmax, backward (dx, dW, db)"""
    g = torch.Generator().manual_seed(C_ + NC)
    B = 3
    x = rq(torch.randn(B, C_, H, W, generator=g), dtype).requires_grad_(True)
    w = (torch.randn(NC, C_, 1, 1, generator=g) / C_ ** 0.5).requires_grad_(True)
    b = torch.randn(NC, generator=g).requires_grad_(True)
    ref = F.conv2d(x, w, b)
    dl = torch.randn(ref.shape, generator=g)
    ref.backward(dl)
    xd = to_nhwc(x.detach(), dtype)
    logits = torch.empty(B, NC, H, W, device=DEV)
    labels = torch.empty(B, H, W, dtype=torch.uint8, device=DEV)
    wd, bd = w.detach().view(NC, C_).contiguous().to(DEV), b.detach().to(DEV)
    ops.linear_head_fwd(ops.view(xd), wd, bd, logits=logits)
    ops.linear_head_fwd(ops.view(xd), wd, bd, labels=labels)
    P = B * H * W
    rows = ops.linear_head_bwd_rows(P)
    part = torch.zeros(rows * (NC * C_ + NC), device=DEV)
    dx = torch.empty_like(xd)
    ops.linear_head_bwd(ops.view(xd), dl.to(DEV), wd, ops.view(dx), part)
    gwb = torch.empty(NC * C_ + NC, device=DEV)
    ops.colsum_finalize(part, rows, NC * C_ + NC, gwb)
    torch.cuda.synchronize()
    close(logits.cpu(), ref.detach(), 1e-5, "logits")
    assert torch.equal(labels.cpu().long(), logits.cpu().argmax(1))
    close(from_nhwc(dx), x.grad, 1e-5 if dtype == torch.float32 else 1e-2, "dx")
    close(gwb[:NC * C_].cpu().view(NC, C_), w.grad.view(NC, C_), 1e-5, "dW")
    close(gwb[NC * C_:].cpu(), b.grad, 1e-5, "db")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("dil,cin,cout,S", [(2, 128, 128, 16), (6, 256, 128, 12), (18, 128, 64, 16), (12, 64, 64, 9)])
def test_dilated_conv_forward_data_and_weight_gradient(dtype, dil, cin, cout, S):
    """atrous 3x3 / stride 1 / pad = dil (ASPP, dilated ResNet stages): forward with statistics and data gradient on the
    generic gather kernel, weight gradient as nine shifted 1x1 problems (GEMM kernel in bf16, generic kernel in f32)"""
    g = torch.Generator().manual_seed(dil)
    B = 2
    x = torch.randn(B, cin, S, S, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    xq, wq = rq(x, dtype).requires_grad_(True), rq(w, dtype).requires_grad_(True)
    ref = F.conv2d(xq, wq, None, 1, dil, dil)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(rq(dy, dtype))
    xd, dyd = to_nhwc(x, dtype), to_nhwc(dy, dtype)
    wf, wdg = ops.pack_conv_weight(w.to(DEV), cin, dtype)
    y = torch.empty(B, S, S, cout, dtype=dtype, device=DEV)
    ops.conv2d(ops.view(xd), wf, None, ops.view(y), 3, 3, 1, dil, dil)
    gx = torch.empty(B, S, S, cin, dtype=dtype, device=DEV)
    ops.conv2d(ops.view(dyd), wdg, None, ops.view(gx), 3, 3, 1, dil, dil)
    dw = torch.empty(cout, cin, 3, 3, device=DEV)
    ws = torch.empty(ops.wgrad_workspace_floats_for(ops.view(xd), ops.view(dyd), 3, 3, 1, dil, dil=dil), device=DEV)
    ops.conv2d_wgrad(ops.view(xd), ops.view(dyd), dw, 3, 3, 1, dil, ws, dil=dil)
    torch.cuda.synchronize()
    t = 1e-4 if dtype == torch.float32 else 2e-2
    close(from_nhwc(y), ref.detach(), t, "forward")
    close(from_nhwc(gx), xq.grad, t, "data gradient")
    close(dw.cpu(), wq.grad, 1e-4, "weight gradient")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,W,a,b", [(8, 12, 1, 2), (8, 12, 2, 1), (16, 16, 2, 4), (16, 16, 4, 1), (12, 12, 1, 3)])
def test_sub_grid_layout_change_and_dilated_conv_as_plain_conv(dtype, H, W, a, b):
    """cvcs_regrid L_a -> L_b against the definition; and the point of it: a 3x3 convolution with dilation d on the map equals the
    plain 3x3 / pad 1 convolution on the d*d sub-sampled images of L_d (forward of the dilated ResNet stages)"""
    g = torch.Generator().manual_seed(H + a + b)
    B, C_ = 2, 32
    x = rq(torch.randn(B, C_, H, W, generator=g), dtype)

    def to_grid(t, d):     # [B, C, H, W] -> L_d as NHWC [B*d*d, H/d, W/d, C]
        Bn, Cn, Hn, Wn = t.shape
        return t.reshape(Bn, Cn, Hn // d, d, Wn // d, d).permute(0, 3, 5, 2, 4, 1).reshape(Bn * d * d, Hn // d, Wn // d, Cn).contiguous()
    src = to_grid(x, a).to(dtype).to(DEV).contiguous()
    out = torch.empty(B * b * b, H // b, W // b, C_, dtype=dtype, device=DEV)
    ops.regrid(ops.view(src), ops.view(out), H, W, a, b)
    torch.cuda.synchronize()
    assert torch.equal(out.float().cpu(), to_grid(x, b))
    if a == 1 and b > 1:
        w = rq(torch.randn(C_, C_, 3, 3, generator=g) / (C_ * 9) ** 0.5, dtype)
        ref = F.conv2d(x, w, None, 1, b, b)
        wf, _ = ops.pack_conv_weight(w.to(DEV), C_, dtype)
        y = torch.empty_like(out)
        ops.conv2d(ops.view(out), wf, None, ops.view(y), 3, 3, 1, 1, 1)
        torch.cuda.synchronize()
        close(y.float().cpu(), to_grid(ref, b), 1e-4 if dtype == torch.float32 else 2e-2, "dilated conv as a plain conv on the sub-grid layout")


# ---------------------------------------------------------------------------------------------------- whole networks
def _build(cls, arch, NC, precision, plus, seed=3, aspp_dropout=0.0):
    net = cls(NC, precision, aspp_dropout=aspp_dropout)      # (0: the deterministic parity runs; the dropout has its own test below)
    missing, unexpected = net.load_state_dict(D.init_params(arch, NC, seed=seed, plus=plus), strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    return net.to(DEV)


def _oracle_named(net):
    """the network's state under the oracle's (= the engine's internal) key names: the reference wrappers DeepLabv3Resnet50/101 export
    torchvision's names (`model.backbone.*`, `model.classifier.*`)"""
    back = getattr(net, "_from_reference", lambda k: k)
    return {back(k): v.detach().cpu().clone() for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}


def test_reference_checkpoint_key_names_of_the_deeplabv3_wrappers(tmp_path):
    """S/nets.py:234-257 + S/utils.py:282-299: a checkpoint as the reference writes it for `Resnet101` (torchvision key names under
    `model.`, a DataParallel `module.` prefix, the unused auxiliary head) loads through `wrapper` / `custom_load`; what this package
    saves carries the same names"""
    net = utils.load_network({"net": "Resnet101", "num_classes": 15, "precision": "bf16"}, DEV)
    assert net.wrapper
    sd = net.state_dict()
    for k in ("model.backbone.conv1.weight", "model.backbone.layer3.22.conv2.weight", "model.classifier.0.convs.3.0.weight",
              "model.classifier.0.convs.4.1.weight", "model.classifier.0.project.1.running_var", "model.classifier.1.weight",
              "model.classifier.2.bias", "model.classifier.4.weight", "model.classifier.4.bias"):
        assert k in sd, k
    assert tuple(sd["model.classifier.4.weight"].shape) == (16, 256, 1, 1)
    ck = {"model_state_dict": {"module." + k: (v.float() * 0 + 0.25 if v.dtype == torch.float32 else v).cpu() for k, v in sd.items()}}
    ck["model_state_dict"]["module.model.aux_classifier.4.weight"] = torch.zeros(21, 256, 1, 1)
    path = str(tmp_path / "checkpoint1")
    torch.save(ck, path)
    other = utils.load_network({"net": "Resnet101", "num_classes": 15, "precision": "bf16"}, DEV)
    utils.load_checkpoint({"load_checkpoint": path}, other)
    out = other.state_dict()
    assert all((v == 0.25).all() for k, v in out.items() if v.dtype == torch.float32 and "aux_classifier" not in k)
    assert (out["model.aux_classifier.4.weight"] == 0).all()          # the unused auxiliary head travels with the checkpoint


def _f64_gradients(arch, p32, img, lab, os_, plus, relu_masks):
    p = {k: v.detach().double().clone() for k, v in p32.items()}
    names = [k for k in p if not D.is_buffer(k)]
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
        loss = O.cross_entropy(D.forward(p, img.double(), arch, train=True, output_stride=os_, plus=plus), lab.long(), None, 0)
    finally:
        F.relu = orig
    return dict(zip(names, torch.autograd.grad(loss, [p[k] for k in names]))), flips


def test_factory_names():
    net = utils.load_network({"net": "DeepLabV3Plus", "num_classes": 4, "precision": "bf16"}, DEV)
    assert isinstance(net, nets.DeepLabV3Plus) and net.returns_logits and not net.requires_context
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")] == \
        [(k, tuple(s)) for k, s in D.param_spec("resnet50", 5, True)]
    r101 = utils.load_network({"net": "Resnet101", "num_classes": 15, "precision": "bf16"}, DEV)   # S/utils.py:180-181
    want = sum(int(torch.tensor(s).prod()) for k, s in D.param_spec("resnet101", 16, False) if not D.is_buffer(k))
    assert isinstance(r101, nets.DeepLabv3Resnet101) and utils.count_params(r101) == want


@pytest.mark.parametrize("cls,arch,os_,plus,B,S,NC", [(nets.DeepLabV3Plus, "resnet50", 16, True, 4, 128, 16),
                                                      (nets.DeepLabv3Resnet50, "resnet50", 8, False, 4, 64, 5)])
def test_fp32_train_steps_match_oracle(cls, arch, os_, plus, B, S, NC):
    """three SGD2 steps (S/train.py:121-126), logits 1e-3 / loss 1e-4 against the f32 oracle; gradients 2e-4 relative L2 per
    tensor against the float64 oracle evaluated at the HIP path's own ReLU decisions, the differing decisions counted and
    each on a pre-activation below 1e-4 of its tensor's max (see tests/test_resnet_gpu.py; the ASPP maps of these small
    tiles are 4x4 .. 16x16 pixels).  Batch 4: the image-pooling branch normalises a 1x1 map over the BATCH - with 2 tiles its
    BatchNorm is so ill-conditioned that torch-CPU f32 itself is 1e-3 from float64 at identical decisions (5e-5 with 4)."""
    net = _build(cls, arch, NC, "fp32", plus)
    tr = D.OracleTrainer(arch, NC, "SGD2", ignore_index=0, seed=3, output_stride=os_, plus=plus)
    crit = utils.CrossEntropyLoss(ignore_index=0)
    optim, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 20}, net)
    img, lab = O.synthetic_tiles(B, S, NC, seed=5, structured=True)
    net.train()
    for step in range(3):
        p_before = _oracle_named(net)
        with torch.no_grad():   # every step is checked at the HIP network's own current parameters
            want_logits = D.forward({k: v.clone() for k, v in p_before.items()}, img.float(), arch, train=True, output_stride=os_, plus=plus)
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
        g64, flips = _f64_gradients(arch, p_before, img, lab, os_, plus, masks)
        assert all(mx <= 1e-4 for _, mx in flips), [f for f in flips if f[0]]
        # the image-pooling branch normalises a 1x1 map over the batch: with B tiles its BatchNorm sees B values per channel,
        # the gradient THROUGH it is a cancellation residue (exactly 0 for B = 2 up to eps) - its conv is held to 5e-3
        # (the stem sits in front of MaxPool2d(3, 2, 1): the oracle is evaluated at the HIP path's ReLU decisions but at its OWN arg-max choices,
        # a near-tie of two window entries re-routes a gradient element of the stem only - its conv and BatchNorm are held to 1e-3)
        def weight(k):
            return 0.04 if k.startswith("aspp.convs.4.") else (0.2 if k.startswith(("encoder.conv1.", "encoder.bn1.")) else 1.0)
        e = sorted((rel_l2(grads[k].double(), g64[k]) * weight(k), k) for k in g64)
        print(f"step {step}: {sum(n for n, _ in flips)} ReLU decisions differ from the float64 oracle's own; gradient rel-L2 vs "
              f"float64 at the HIP decisions: worst {e[-1]}, median {e[len(e) // 2][0]:.2e}")
        assert e[-1][0] <= 2e-4, e[-1]
        if step == 0:   # the first update against the oracle trainer (identical parameters going in)
            tr.step(img, lab)
            sd = _oracle_named(net)
            worstp = max((rel_l2(sd[k], tr.p[k].detach()), k) for k in tr.p)
            # (BatchNorm biases start at 0: their relative figure after one update IS the relative error between two f32 gradients,
            #  each with its own handful of ReLU decisions - the same 2e-4 as the gradient bound above; weights are 100x tighter)
            assert worstp[0] <= 2e-4, worstp
    net.eval()
    final = _oracle_named(net)
    with torch.no_grad():
        ev = net(img.to(DEV), None).cpu()
        want = D.forward(final, img.float(), arch, train=False, output_stride=os_, plus=plus)
        labels = net.predict_labels(img.to(DEV)).cpu()
    close(ev, want, 1e-3, "eval logits")
    assert torch.equal(labels.long(), ev.argmax(1))


def test_bf16_layer_by_layer_and_end_to_end():
    """DeepLabV3+ at the benchmarked precision, 2 x 256 x 256: every conv output recomputed in f32 from the path's own stored
    bf16 input must agree to bf16 storage rounding; end to end within the rounding noise floor of this network (the distance
    between the f32 oracle and its own bf16 emulation) and 1e-2 RMS of max|logit|"""
    arch, NC, B, S = "resnet50", 16, 4, 256
    net = _build(nets.DeepLabV3Plus, arch, NC, "bf16", True)
    p = D.init_params(arch, NC, seed=3)
    img, lab = O.synthetic_tiles(B, S, NC, seed=9, structured=True)
    net.train()
    logits = net(img.to(DEV), None)
    torch.cuda.synchronize()
    eng = net._engine
    worst = (0.0, "")
    from test_resnet_gpu import check_gram_tail
    for conv, (u, act_out) in eng.units.items():
        if getattr(u, "gram", False):      # a bottleneck tail without a stored conv3 output: statistics + block output from the stored operands
            worst = max(worst, (check_gram_tail(eng, u), conv))
            continue
        x = u.x.torch().float().cpu().permute(0, 3, 1, 2)
        if u.virt:
            x = x[:, :3, :, 3:-5]
        w = eng.P[conv + ".weight"].detach().cpu().to(torch.bfloat16).float()
        ref = F.conv2d(x, w, None, 2 if u.virt else u.stride, 3 if u.virt else u.pad, u.dil)
        got = from_nhwc(u.y.torch())
        e = (got - ref).abs().max().item() / ref.abs().max().item()
        worst = max(worst, (e, conv))
        assert e <= 2.0 ** -8, f"{conv}: conv output off by {e:.3e} of its max"
    print(f"DeepLabV3+ {B}x{S}: worst conv output error {worst[0]:.3e} ({worst[1]}) of its max [bf16 ulp = 3.9e-3]")
    with torch.no_grad():
        want = D.forward({k: v.clone() for k, v in p.items()}, img.float(), arch, train=True, emulate_bf16=True)
        want32 = D.forward({k: v.clone() for k, v in p.items()}, img.float(), arch, train=True)
    got = logits.detach().cpu()
    scale = want32.abs().max().item()
    floor, floor_rms = (want - want32).abs().max().item() / scale, (want - want32).pow(2).mean().sqrt().item() / scale
    e32, rms = (got - want32).abs().max().item() / scale, (got - want32).pow(2).mean().sqrt().item() / scale
    print(f"bf16 logits / max|logit|: max {e32:.3e}, rms {rms:.3e} from the f32 oracle; emulation vs f32 oracle max {floor:.3e}, "
          f"rms {floor_rms:.3e}")
    # (the image-pooling BatchNorm over B = 4 values per channel makes this network's rounding noise floor high: both bounds
    #  are relative to the floor measured on the CPU)
    assert e32 <= 1.5 * floor and rms <= 1.5 * floor_rms


def test_deeplab_trains_reproducibly():
    img, lab = O.synthetic_tiles(4, 128, 5, seed=2, structured=True)
    runs = []
    for _ in range(2):
        net = _build(nets.DeepLabV3Plus, "resnet50", 5, "bf16", True)
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
    assert runs[0][0][-1] < 0.7 * runs[0][0][0], runs[0][0]


def test_aspp_dropout_mask_is_counter_based_and_the_train_step_matches_the_oracle_under_the_replayed_mask():
    """The reference's `Resnet101` trains with torchvision's nn.Dropout(0.5) behind the ASPP projection (S/nets.py:242-246; S/train.py:113).
    Here the mask is a counter-based hash of (seed, step, element): (1) the kernel against the host replay, bit for bit, forward = backward
    mask, ~half kept; (2) two f32 train steps of DeepLabV3 (OS 8, the wrapper's key names) against the oracle given the REPLAYED masks of
    steps 0 and 1: logits 1e-3, loss 1e-4, and the steps draw different masks; eval mode applies none."""
    from cvcs_amd import ops
    B, H, W, C = 2, 5, 7, 64
    x = torch.randn(B, H, W, C, device=DEV)
    state = torch.tensor([1234, 3], dtype=torch.int64, device=DEV)
    out = torch.empty_like(x)
    ops.dropout(ops.view(x), ops.view(out), state, 0.5)
    keep = ops.dropout_mask_host(1234, 3, x.numel(), 0.5).view(B, H, W, C)
    assert torch.equal(out.cpu(), torch.where(keep, x.cpu() * 2.0, torch.zeros(())))
    assert 0.45 < keep.float().mean().item() < 0.55
    ops.counter_add(state, 1)
    torch.cuda.synchronize()
    assert state.tolist() == [1234, 4]
    # the network
    NC, S, seed = 5, 64, 77
    arch, os_ = "resnet50", 8
    net = nets.DeepLabv3Resnet50(NC, "fp32", dropout_seed=seed)          # class default: p = 0.5
    assert net.aspp_dropout == 0.5
    net.load_state_dict({net._to_reference(k): v for k, v in D.init_params(arch, NC, seed=3, plus=False).items()}, strict=False)
    net = net.to(DEV)
    tr = D.OracleTrainer(arch, NC, "SGD2", ignore_index=0, seed=3, output_stride=os_, plus=False)
    img, lab = O.synthetic_tiles(4, S, NC, seed=5, structured=True)
    crit = utils.CrossEntropyLoss(ignore_index=0)
    opt, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 20}, net)
    net.train()
    h5 = S // os_
    masks = []
    for step in range(2):
        keep = ops.dropout_mask_host(seed, step, 4 * h5 * h5 * 256, 0.5).view(4, h5, h5, 256).permute(0, 3, 1, 2)
        masks.append(keep)
        logits = net(img.to(DEV), None)
        loss = crit(logits, lab.to(DEV))
        opt.zero_grad()
        loss.backward()
        opt.step()
        # the oracle, same step, the replayed mask as an input
        z = D.forward(tr.p, img.float(), arch, train=True, output_stride=os_, plus=False, aspp_keep=keep)
        lo = O.cross_entropy(z, lab.long(), None, 0)
        grads = torch.autograd.grad(lo, [tr.p[k] for k in tr.trainable])
        scale = z.abs().max().item()
        err = (logits.detach().cpu() - z.detach()).abs().max().item() / scale
        # step 0 is the parity statement; after one optimiser step this small network has amplified the f32 rounding differences of
        # step 0 (the exact-mode test measures the same chaos: a permuted batch lands 2 % away): step 1 only has to tell the RIGHT mask
        # from a wrong one, which is an O(1) error
        etol, ltol = (1e-3, 1e-4) if step == 0 else (5e-2, 2e-3)
        assert err <= etol and abs(loss.item() - lo.item()) <= ltol * max(1.0, abs(lo.item())), (step, err, loss.item(), lo.item())
        if step == 1:
            with torch.no_grad():
                zw = D.forward({k: v.detach().clone() for k, v in tr.p.items()}, img.float(), arch, train=True, output_stride=os_, plus=False, aspp_keep=masks[0])
            assert (logits.detach().cpu() - zw).abs().max().item() / scale > 5 * max(err, 1e-3)
        tr.nstep += 1
        with torch.no_grad():
            for k, g in zip(tr.trainable, grads):
                newp, buf = O.sgd_step(tr.p[k], g, tr.state.get(k), tr.lr(), tr.cfg["momentum"], tr.cfg["weight_decay"], k not in tr.state)
                tr.state[k] = buf
                tr.p[k].copy_(newp)
    assert not torch.equal(masks[0], masks[1])
    assert net._engine.drop_state.tolist() == [seed, 2]
    net.eval()
    with torch.no_grad():
        ev = net(img.to(DEV), None).cpu()
        # (the oracle on THIS network's weights after its two steps: the statement is "evaluation applies no mask", not a third parity step)
        want = D.forward(_oracle_named(net), img.float(), arch, train=False, output_stride=os_, plus=False)
    assert (ev - want).abs().max().item() <= 2e-3 * want.abs().max().item()


def test_reference_checkpoint_with_the_aux_head_survives_a_round_trip(tmp_path):
    """a checkpoint of the reference's `Resnet101` holds torchvision's auxiliary FCN head (`model.aux_classifier.{0,1,4}.*`, built with the
    COCO weights; unused by its forward, S/nets.py:248-250): loading it here and saving again must give back EVERY key, so the reference's
    strict `custom_load` (S/nets.py:252-257) takes the result"""
    net = utils.load_network({"net": "Resnet101", "num_classes": 15, "precision": "bf16"}, DEV)
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    aux = {"model.aux_classifier.0.weight": torch.randn(256, 1024, 3, 3), "model.aux_classifier.1.weight": torch.ones(256),
           "model.aux_classifier.1.bias": torch.zeros(256), "model.aux_classifier.1.running_mean": torch.zeros(256),
           "model.aux_classifier.1.running_var": torch.ones(256), "model.aux_classifier.1.num_batches_tracked": torch.tensor(0),
           "model.aux_classifier.4.weight": torch.randn(21, 256, 1, 1), "model.aux_classifier.4.bias": torch.zeros(21)}
    assert not any(k in sd for k in aux)              # a fresh network has no aux head (documented: strict=False on the reference side)
    ref_ck = {"module." + k: v for k, v in {**sd, **aux}.items()}
    other = utils.load_network({"net": "Resnet101", "num_classes": 15, "precision": "bf16"}, DEV)
    other.custom_load({"model_state_dict": ref_ck})
    back = other.state_dict()
    assert set(back) == set(sd) | set(aux)
    for k, v in aux.items():
        assert torch.equal(back[k].cpu(), v), k
