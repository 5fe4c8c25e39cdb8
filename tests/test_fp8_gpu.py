"""fp8 convolution path (BASELINE configs[4] "mixed bf16/fp8 convs") on the MI355X against the oracle's fp8 emulation:
the quantiser bit for bit, the block-scaled-MFMA 3x3 convolution against a float64 convolution of the SAME fp8 operands."""
import pytest
import torch

from oracle import fp8_emulation as F8

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _view(t, off=0, c=None):
    from cvcs_amd import ops
    return ops.view(t, off, c)


@pytest.mark.parametrize("fmt", [F8.E4M3, F8.E5M2])
def test_quantizer_matches_the_emulation_bit_for_bit(fmt):
    from cvcs_amd import ops
    g = torch.Generator().manual_seed(3 + fmt)
    B, H, W, C, ld = 2, 9, 13, 48, 80
    x = (torch.randn(B, H, W, ld, generator=g) * torch.exp(torch.randn(B, H, W, ld, generator=g) * 3)).bfloat16()
    x[0, 0, 0, 5] = 0.0
    xd = x.to(DEV)
    q = torch.zeros(B, H, W, 64, dtype=torch.uint8, device=DEV)
    slots = ops.Fp8Slots(DEV)
    i = slots.slot("x", fmt)
    xv, qv = _view(xd, 16, C), _view(q, 0, C)
    ops.quantize_fp8(xv, qv, fmt, slots, i)          # first use: calibrates on its own amax, then quantises
    torch.cuda.synchronize()
    part = x[..., 16:16 + C].float()
    amax = part.abs().max().item()
    st = slots.t[i].cpu()
    assert slots.amax_now(i) == amax                  # this step's amax (the recorded launch re-took it)
    scale = F8.scale_from_amax(amax, fmt)
    assert st[1].item() == scale and st[3].item() == F8.FMAX[fmt]
    inv = torch.tensor(1.0) / torch.tensor(scale)
    want = F8.quantize_bits(part * inv, fmt)
    got = q[..., :C].cpu()
    assert torch.equal(got, want), f"{(got != want).sum().item()} of {want.numel()} codes differ"
    assert int(q[..., C:].abs().sum()) == 0           # bytes outside the view untouched
    # delayed scaling: the next step scales by THIS step's amax, whatever the tensor then holds
    slots.update()
    x2 = (xd * 0.25).contiguous()
    ops.quantize_fp8(_view(x2, 16, C), qv, fmt, slots, i)
    torch.cuda.synchronize()
    assert slots.t[i, 1].item() == scale
    want2 = F8.quantize_bits(x2[..., 16:16 + C].cpu().float() * inv, fmt)
    assert torch.equal(q[..., :C].cpu(), want2)
    assert slots.amax_now(i) == x2[..., 16:16 + C].float().abs().max().item()


CONV_CASES = [  # B, H, W, Cin, Cout, fmt, in_ld_extra, out_ld_extra
    (2, 16, 16, 128, 128, F8.E4M3, 0, 0),
    (1, 32, 48, 256, 128, F8.E4M3, 128, 64),      # two slices, views with a leading dimension (concat buffers)
    (2, 24, 40, 384, 256, F8.E4M3, 0, 0),         # ragged tiles, odd slice count, two column tiles
    (1, 8, 8, 128, 128, F8.E5M2, 0, 0),           # the smallest map; e5m2 pixels (data gradient)
    (2, 20, 36, 512, 128, F8.E5M2, 0, 0),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_fp8_conv3x3_against_float64_on_the_same_operands(case):
    from cvcs_amd import ops
    B, H, W, Cin, Cout, fmt, xe, oe = case
    g = torch.Generator().manual_seed(Cin + Cout + H)
    slots = ops.Fp8Slots(DEV)
    ix, iw = slots.slot("x", fmt), slots.slot("w", F8.E4M3)
    x = (torch.randn(B, H, W, Cin + xe, generator=g) * 1.5).bfloat16().to(DEV)
    w = (torch.randn(9, Cout, Cin, generator=g) * 0.05).bfloat16().to(DEV)
    x8 = torch.zeros(B, H, W, Cin + xe, dtype=torch.uint8, device=DEV)
    w8 = torch.zeros(9, Cout, Cin, dtype=torch.uint8, device=DEV)
    xv = _view(x, xe, Cin)
    ops.quantize_fp8(xv, _view(x8, xe, Cin), fmt, slots, ix)
    wv = ops.View(w.view(1, 9 * Cout, 1, Cin), 0, Cin)
    ops.quantize_fp8(wv, ops.View(w8.view(1, 9 * Cout, 1, Cin), 0, Cin), F8.E4M3, slots, iw)
    out = torch.full((B, H, W, Cout + oe), 7.0, dtype=torch.bfloat16, device=DEV)
    rows = ops.fp8_stat_rows(xv)
    stats = tuple(torch.zeros(n, dtype=torch.float32, device=DEV) for n in (rows * Cout, rows * Cout, rows))
    ops.conv3x3_fp8(_view(x8, xe, Cin), fmt, w8, _view(out, oe, Cout), slots, ix, iw, stats=stats)
    torch.cuda.synchronize()
    sx, sw = slots.t[ix, 1].item(), slots.t[iw, 1].item()
    xf = F8.dequantize_bits(x8[..., xe:].cpu(), fmt).double().permute(0, 3, 1, 2)
    wf = F8.dequantize_bits(w8.cpu(), F8.E4M3).double().view(3, 3, Cout, Cin).permute(2, 3, 0, 1)
    ref = torch.nn.functional.conv2d(xf, wf, padding=1) * (sx * sw)
    got = out[..., oe:].float().cpu().permute(0, 3, 1, 2).double()
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item() / scale
    assert err < 6e-3, f"fp8 conv differs from float64 on its own operands by {err:.2e} of max|y| (bf16 ulp 3.9e-3)"
    assert float(out[..., :oe].float().min()) == 7.0 if oe else True      # channels outside the view untouched
    # BatchNorm partial statistics of the values as stored
    ssum, sm2, scnt = (t.cpu().double() for t in stats)
    assert scnt.sum().item() == B * H * W
    tot = ssum.view(rows, Cout).sum(0)
    want = got.sum(dim=(0, 2, 3))
    assert (tot - want).abs().max().item() <= 2e-3 * got.abs().sum(dim=(0, 2, 3)).max().item() + 1e-3
    # merged second moment (Chan) against the variance of the stored values
    n_r = scnt.view(rows, 1)
    mean_r = ssum.view(rows, Cout) / n_r
    mean = tot / (B * H * W)
    m2 = sm2.view(rows, Cout).sum(0) + (n_r * (mean_r - mean) ** 2).sum(0)
    want_m2 = ((got - got.mean(dim=(0, 2, 3), keepdim=True)) ** 2).sum(dim=(0, 2, 3))
    assert ((m2 - want_m2).abs() / want_m2.clamp_min(1e-6)).max().item() < 2e-2


def test_fp8_conv3x3_eval_fold_and_relu():
    from cvcs_amd import ops
    B, H, W, Cin, Cout = 1, 16, 32, 128, 128
    g = torch.Generator().manual_seed(5)
    slots = ops.Fp8Slots(DEV)
    ix, iw = slots.slot("x", F8.E4M3), slots.slot("w", F8.E4M3)
    x = torch.randn(B, H, W, Cin, generator=g).bfloat16().to(DEV)
    w = (torch.randn(9, Cout, Cin, generator=g) * 0.05).bfloat16().to(DEV)
    x8, w8 = torch.zeros_like(x, dtype=torch.uint8), torch.zeros_like(w, dtype=torch.uint8)
    ops.quantize_fp8(_view(x), _view(x8), F8.E4M3, slots, ix)
    ops.quantize_fp8(ops.View(w.view(1, 9 * Cout, 1, Cin), 0, Cin), ops.View(w8.view(1, 9 * Cout, 1, Cin), 0, Cin), F8.E4M3, slots, iw)
    ps = (torch.rand(Cout, generator=g) + 0.5).to(DEV)
    pt = torch.randn(Cout, generator=g).to(DEV)
    out = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=DEV)
    ops.conv3x3_fp8(_view(x8), F8.E4M3, w8, _view(out), slots, ix, iw, relu=True, pre_affine=(ps, pt))
    torch.cuda.synchronize()
    sx, sw = slots.t[ix, 1].item(), slots.t[iw, 1].item()
    xf = F8.dequantize_bits(x8.cpu(), F8.E4M3).double().permute(0, 3, 1, 2)
    wf = F8.dequantize_bits(w8.cpu(), F8.E4M3).double().view(3, 3, Cout, Cin).permute(2, 3, 0, 1)
    ref = torch.relu(torch.nn.functional.conv2d(xf, wf, padding=1) * (sx * sw) * ps.cpu().double().view(1, -1, 1, 1) + pt.cpu().double().view(1, -1, 1, 1))
    got = out.float().cpu().permute(0, 3, 1, 2).double()
    assert (got - ref).abs().max().item() / ref.abs().max().item() < 6e-3


# ------------------------------------------------------------------------------------------------ the fp8 network path
def _build_fp8(NC, seed=3, precision="fp8"):
    from cvcs_amd import nets
    from oracle import resnet_unet_oracle as R
    net = nets.Resnet50Unet(NC, precision)
    missing, unexpected = net.load_state_dict(R.init_params("resnet50", NC, seed=seed), strict=False)
    assert not unexpected
    return net.to(DEV)


@pytest.mark.parametrize("NC,B,S", [(16, 2, 256), (21, 1, 1024)], ids=["2x256x256", "cfg5 as named: 1024x1024, 20 classes"])
def test_fp8_network_layer_by_layer_forward_and_backward(NC, B, S):
    """precision "fp8" of ResNet50-UNet (BASELINE configs[4]: "ResNet50-UNet 1024x1024 tiles, 20 classes, mixed bf16/fp8 convs") at 2 x 256 x 256
    and AS NAMED at 1024 x 1024 with 20 + 1 classes, one train step, LAYER BY LAYER from the path's OWN stored
    operands (no error amplification): (1) which layers run fp8 = the oracle's eligibility rule; (2) every fp8 image is the emulation's
    quantisation of the bf16 tensor it was taken from, bit for bit, at the slot's scale, and the scale is 2 amax / fmax of that tensor
    (first step: calibrated on itself); (3) every fp8 conv output / data gradient equals a float64 convolution of the stored fp8 operands
    to bf16 storage rounding."""
    from cvcs_amd import utils
    from oracle import resnet_unet_oracle as R
    from oracle import unet_oracle as O
    net = _build_fp8(NC)
    net._ensure_flat()
    eng = net._engine
    eng.keep_all = True
    img, lab = O.synthetic_tiles(B, S, NC, seed=9, structured=True)
    net.train()
    crit = utils.CrossEntropyLoss(ignore_index=0)
    loss = crit(net(img.to(DEV), None), lab.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    assert loss.item() == loss.item()
    slots = eng.f8
    table = slots.t.cpu()
    bufs = {k[0]: v for k, v in eng._bufs.items()}
    p = R.init_params("resnet50", NC, seed=3)
    n8 = 0
    worst = {"fwd": 0.0, "dgrad": 0.0}
    for conv, (u, act_out) in eng.units.items():
        w = p[conv + ".weight"]
        xs = (u.x.B, u.x.C, u.x.H, u.x.W)
        want = R.fp8_eligible(w, torch.empty(xs, device="meta"), u.stride, u.pad, u.dil) and not u.virt
        assert bool(getattr(u, "fp8", False)) == bool(want), conv
        if not want:
            continue
        n8 += 1
        ia, iw, ig = slots.names["a:" + conv], slots.names["w:" + conv], slots.names["g:" + conv]
        # (2) quantisation of the stored bf16 input / packed weights / gradient
        x_bf = u.x.torch().float().cpu()
        sx = table[ia, 1].item()
        assert sx == F8.scale_from_amax(x_bf.abs().max().item(), F8.E4M3), conv
        x8 = bufs["f8.a:" + conv][..., u.x.off:u.x.off + u.x.C].cpu()
        assert torch.equal(x8, F8.quantize_bits(x_bf * (torch.tensor(1.0) / torch.tensor(sx)), F8.E4M3)), conv + ": input image"
        wf = eng.packed[conv]["wf"].float().cpu()
        sw = table[iw, 1].item()
        assert sw == F8.scale_from_amax(wf.abs().max().item(), F8.E4M3)
        wf8 = eng.packed[conv]["wf8"].cpu()
        assert torch.equal(wf8, F8.quantize_bits(wf * (torch.tensor(1.0) / torch.tensor(sw)), F8.E4M3)), conv + ": weight image"
        # (3) forward: float64 convolution of the fp8 operands
        xf = F8.dequantize_bits(x8, F8.E4M3).double().permute(0, 3, 1, 2)
        wq = F8.dequantize_bits(wf8, F8.E4M3).double().view(3, 3, w.shape[0], w.shape[1]).permute(2, 3, 0, 1)
        ref = torch.nn.functional.conv2d(xf, wq, padding=1) * (sx * sw)
        got = u.y.torch().float().cpu().permute(0, 3, 1, 2).double()
        e = (got - ref).abs().max().item() / ref.abs().max().item()
        worst["fwd"] = max(worst["fwd"], e)
        assert e <= 2.0 ** -7, f"{conv}: fp8 conv output off by {e:.3e} of its max"
        # data gradient: e5m2 image of the stored dy, the SAME weight values in the flipped layout
        r = eng.bwd_units[conv]
        dy_bf = r["dy"].torch().float().cpu()
        sg = table[ig, 1].item()
        assert sg == F8.scale_from_amax(dy_bf.abs().max().item(), F8.E5M2), conv
        dy8 = bufs["f8.g:" + conv].cpu()
        assert torch.equal(dy8, F8.quantize_bits(dy_bf * (torch.tensor(1.0) / torch.tensor(sg)), F8.E5M2)), conv + ": gradient image"
        gq = F8.dequantize_bits(dy8, F8.E5M2).double().permute(0, 3, 1, 2)
        refg = torch.nn.grad.conv2d_input(xf.shape, wq, gq, padding=1) * (sg * sw)
        gotg = r["gx"].torch().float().cpu().permute(0, 3, 1, 2).double()
        eg = (gotg - refg).abs().max().item() / refg.abs().max().item()
        worst["dgrad"] = max(worst["dgrad"], eg)
        assert eg <= 2.0 ** -7, f"{conv}: fp8 data gradient off by {eg:.3e} of its max"
    assert n8 == 14, n8      # layer2 (3) + layer3 (5) + layer4 (2) stride-1 3x3s, four decoder convs
    print(f"fp8 layers: {n8}; worst forward error {worst['fwd']:.2e}, worst data-gradient error {worst['dgrad']:.2e} of the tensor's max (bf16 ulp 3.9e-3)")


def test_fp8_network_against_the_emulating_oracle_and_the_bf16_path():
    """End to end, three SGD2 steps at 2 x 256 x 256 (train mode): the fp8 path against the oracle's fp8 emulation (same quantisation
    points, delayed scaling) - inside the rounding-noise floor of the network, measured as the distance between the oracle's bf16 and
    fp8 emulations and its f32 run - and its loss beside the bf16 path's."""
    from cvcs_amd import utils
    from oracle import resnet_unet_oracle as R
    from oracle import unet_oracle as O
    NC, B, S = 16, 2, 256
    img, lab = O.synthetic_tiles(B, S, NC, seed=9, structured=True)
    runs = {}
    for prec in ("fp8", "bf16"):
        net = _build_fp8(NC, precision=prec)
        crit = utils.CrossEntropyLoss(ignore_index=0)
        opt, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 20}, net)
        net.train()
        out = []
        for _ in range(3):
            logits = net(img.to(DEV), None)
            loss = crit(logits, lab.to(DEV))
            out.append((loss.item(), logits.detach().cpu().clone()))
            opt.zero_grad()
            loss.backward()
            opt.step()
        runs[prec] = out
    tr8 = R.OracleTrainer("resnet50", NC, opt="SGD2", ignore_index=0, seed=3, emulate_fp8=True)
    tr16 = R.OracleTrainer("resnet50", NC, opt="SGD2", ignore_index=0, seed=3, emulate_bf16=True)
    tr32 = R.OracleTrainer("resnet50", NC, opt="SGD2", ignore_index=0, seed=3)
    for step in range(3):
        l8, z8, _ = tr8.step(img, lab)
        l16, z16, _ = tr16.step(img, lab)
        l32, z32, _ = tr32.step(img, lab)
        scale = z32.abs().max().item()
        floor = max((z8 - z32).abs().max().item(), (z16 - z32).abs().max().item()) / scale
        got = runs["fp8"][step][1]
        e_emul = (got - z8).abs().max().item() / scale
        e_f32 = (got - z32).abs().max().item() / scale
        rms = (got - z8).pow(2).mean().sqrt().item() / scale
        rms32 = (got - z32).pow(2).mean().sqrt().item() / scale
        floor_rms = max((z8 - z32).pow(2).mean().sqrt().item(), (z16 - z32).pow(2).mean().sqrt().item()) / scale
        print(f"step {step}: loss fp8 HIP {runs['fp8'][step][0]:.5f} | fp8 oracle {l8:.5f} | bf16 HIP {runs['bf16'][step][0]:.5f} | bf16 oracle {l16:.5f} | f32 oracle {l32:.5f}; "
              f"logits / max|logit|: {e_emul:.2e} from the fp8 oracle (rms {rms:.2e}), {e_f32:.2e} from f32; emulation floor {floor:.2e} (rms {floor_rms:.2e})")
        # two realisations of the same rounding noise are sqrt(2) floors apart (and the max norm is one pixel's luck on top: 2 x); against the
        # noiseless f32 run the path shows ONE floor
        assert e_emul <= 2.0 * floor and e_f32 <= 1.5 * floor, (step, e_emul, e_f32, floor)
        assert rms32 <= 1.5 * floor_rms and rms <= 2.0 * floor_rms, (step, rms, rms32, floor_rms)
        assert abs(runs["fp8"][step][0] - l8) <= 1e-2 * max(1.0, abs(l8))
        assert abs(runs["fp8"][step][0] - runs["bf16"][step][0]) <= 2e-2 * max(1.0, abs(l8))
    assert runs["fp8"][2][0] < runs["fp8"][0][0]


def test_fp8_training_is_bitwise_reproducible_and_eval_uses_the_trained_scales():
    from cvcs_amd import utils
    from oracle import unet_oracle as O
    NC, B, S = 5, 2, 128
    img, lab = O.synthetic_tiles(B, S, NC, seed=2, structured=True)
    finals = []
    for _ in range(2):
        net = _build_fp8(NC)
        crit = utils.CrossEntropyLoss(ignore_index=0)
        opt, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 20}, net)
        net.train()
        losses = []
        for _ in range(4):
            loss = crit(net(img.to(DEV), None), lab.to(DEV))
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.item())
        scales = net._engine.f8.t[:, 1].clone()
        net.eval()
        with torch.no_grad():
            ev = net(img.to(DEV), None).clone()
            labels = net.predict_labels(img.to(DEV)).clone()
        assert torch.equal(net._engine.f8.t[:, 1], scales)          # an evaluation pass neither moves the scales ...
        assert float(net._engine.f8.t[:, 16::16].abs().max()) == 0.0   # ... nor feeds the next step's amax
        assert torch.equal(labels.long(), ev.argmax(1))
        finals.append((losses, net.flat_parameters()[0].clone(), ev))
    assert finals[0][0] == finals[1][0] and torch.equal(finals[0][1], finals[1][1]) and torch.equal(finals[0][2], finals[1][2])
    assert all(v == v for v in finals[0][0]) and finals[0][0][-1] < finals[0][0][0]


def test_fp8_on_the_upernet_head_of_swin():
    """`precision: fp8` on Swin-T + UPerNet: the head's 3x3 convolutions whose maps are at least 8 pixels wide (bottleneck, FPN convs, the 2048 -> 512
    fusion conv) run forward and data gradient on the fp8 kernel, the transformer and every weight gradient stay bf16.  Three SGD2 steps: the fp8
    launches really happen, losses follow the bf16 run to 2 %, logits stay within 4 x the bf16 run's own distance from the f32 oracle (measured 2.2 x in RMS: five
    e4m3 / e5m2 convolutions of K = 4608 ... 25 344 on top of the bf16 network's noise), the run is bitwise reproducible."""
    from cvcs_amd import nets, utils
    from oracle import swin_upernet_oracle as W
    from oracle import unet_oracle as O
    NC, B, S = 6, 4, 256
    img, lab = O.synthetic_tiles(B, S, NC, seed=21, structured=True)
    p0 = W.init_params(NC, seed=5)

    def run(precision):
        net = nets.SwinTUperNet(NC, precision)
        missing, unexpected = net.load_state_dict({k: v.clone() for k, v in p0.items()}, strict=False)
        assert not unexpected
        net = net.to(DEV)
        crit = utils.CrossEntropyLoss(ignore_index=0)
        opt, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 20}, net)
        net.train()
        out = []
        for _ in range(3):
            logits = net(img.to(DEV), None)
            loss = crit(logits, lab.to(DEV))
            opt.zero_grad()
            loss.backward()
            opt.step()
            out.append((loss.item(), logits.detach().float().cpu().clone()))
        return net, out

    n8, r8 = run("fp8")
    _, r8b = run("fp8")
    _, r16 = run("bf16")
    eng = n8._engine
    used = sorted(k for k, (u, _) in eng.units.items() if getattr(u, "fp8", False))
    assert "decode_head.fpn_bottleneck.conv" in used and "decode_head.bottleneck.conv" in used and len(used) == 5, used
    assert all(torch.equal(a[1], b[1]) and a[0] == b[0] for a, b in zip(r8, r8b)), "fp8 training must be bitwise reproducible"
    tr = W.OracleTrainer(NC, opt="SGD2", ignore_index=0, seed=5)
    for step in range(3):
        lo, z, _ = tr.step(img, lab)
        scale = z.abs().max().item()
        e16 = (r16[step][1] - z).abs().max().item() / scale
        e8 = (r8[step][1] - z).abs().max().item() / scale
        rms16 = (r16[step][1] - z).pow(2).mean().sqrt().item() / scale
        rms8 = (r8[step][1] - z).pow(2).mean().sqrt().item() / scale
        print(f"step {step}: loss fp8 {r8[step][0]:.5f} bf16 {r16[step][0]:.5f} f32 oracle {lo:.5f}; logits / max|logit| from f32: fp8 {e8:.2e} (rms {rms8:.2e}), bf16 {e16:.2e} (rms {rms16:.2e})")
        assert abs(r8[step][0] - r16[step][0]) <= 2e-2 * max(1.0, abs(lo))
        assert rms8 <= 4.0 * rms16 + 1e-3 and e8 <= 4.0 * e16 + 1e-2, (step, e8, e16, rms8, rms16)
    assert r8[2][0] < r8[0][0]


@pytest.mark.parametrize("variant", ["Unetv2", "Unet"])
def test_fp8_on_the_reference_unets(variant):
    """`precision: fp8` on the reference's own U-Nets (S/nets.py:35-199): the 3x3 convolutions of levels 2-5 and of their decoder stages (both channel
    counts multiples of 128) run forward and data gradient on the fp8 kernel - the conv bias rides in the epilogue's per-channel affine, the
    decoder's conv -> ReLU -> BatchNorm order takes its statistics from the ReLU output.  Three SGD2 steps against the bf16 run and the f32 oracle."""
    from cvcs_amd import nets, utils
    from oracle import unet_oracle as O
    NC, B, S = 6, 4, 128
    img, lab = O.synthetic_tiles(B, S, NC, seed=31, structured=True)
    p0 = O.init_params(variant, NC, seed=7)

    def run(precision):
        net = (nets.Urnetv2 if variant == "Unetv2" else nets.Urnet)(NC, precision)
        net.load_state_dict({k: v.clone() for k, v in p0.items()}, strict=False)
        net = net.to(DEV)
        crit = utils.CrossEntropyLoss(ignore_index=0)
        opt, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 20}, net)
        net.train()
        out = []
        for _ in range(3):
            logits = net(img.to(DEV), None)
            loss = crit(logits, lab.to(DEV))
            opt.zero_grad()
            loss.backward()
            opt.step()
            out.append((loss.item(), logits.detach().float().cpu().clone()))
        return net, out

    n8, r8 = run("fp8")
    _, r8b = run("fp8")
    _, r16 = run("bf16")
    names = set(n8._engine.f8.names)
    fwd = sorted(k for k in names if k.startswith("a:"))
    bwd = sorted(k for k in names if k.startswith("g:"))
    assert len(fwd) >= 12 and len(bwd) >= 12 and not any("encode1." in k or "decode_forward4" in k for k in fwd), (fwd, bwd)
    assert all(torch.equal(a[1], b[1]) and a[0] == b[0] for a, b in zip(r8, r8b)), "fp8 training must be bitwise reproducible"
    tr = O.OracleTrainer(variant, NC, opt="SGD2", ignore_index=0, seed=7)
    for step in range(3):
        lo, z, _ = tr.step(img, lab)
        scale = z.abs().max().item()
        e16 = (r16[step][1] - z).abs().max().item() / scale
        e8 = (r8[step][1] - z).abs().max().item() / scale
        rms16 = (r16[step][1] - z).pow(2).mean().sqrt().item() / scale
        rms8 = (r8[step][1] - z).pow(2).mean().sqrt().item() / scale
        print(f"{variant} step {step}: loss fp8 {r8[step][0]:.5f} bf16 {r16[step][0]:.5f} f32 oracle {lo:.5f}; logits / max|logit| from f32: fp8 {e8:.2e} (rms {rms8:.2e}), bf16 {e16:.2e} (rms {rms16:.2e})")
        # (these 4 x 128^2 random-init runs are steep - the bilinear `Unet` drops from 2.79 to 1.44 in ONE step - and amplify any rounding: the
        #  trajectory bounds are loose by necessity; the statement about the fp8 path is the layer-by-layer check below)
        assert abs(r8[step][0] - r16[step][0]) <= (5e-2 if variant == "Unetv2" else 1e-1) * max(1.0, abs(lo))
        assert rms8 <= 4.0 * rms16 + 2e-3 and e8 <= 4.0 * e16 + 2e-2, (step, e8, e16, rms8, rms16)
    assert r8[2][0] < r8[0][0]
    # the plumbing, layer by layer from the path's own stored operands (one more train-mode forward; no backward, so the scales are the ones
    # the images were taken with): y = [relu](s_x s_w conv(x8, w8) + bias) in float64 from the fp8 codes
    from oracle import fp8_emulation as E
    eng = n8._engine
    n8.train()
    with torch.no_grad():
        n8(img.to(DEV), None)
    torch.cuda.synchronize()
    import torch.nn.functional as F
    checks = [("encode3.1.layer.0", eng.pool[2], eng.ya[3], False), ("decode_forward2.0.layer.0", eng.cat[3], eng.ra[2], True)]
    for conv, xin, y, relu in checks:
        pk = eng.packed[conv]
        ia, iw = eng.f8.names["a:" + conv], eng.f8.names["w:" + conv]
        sx, sw = eng.f8.scale(ia), eng.f8.scale(iw)
        x8 = eng._q8buf[(xin.data_ptr(), tuple(xin.shape))].cpu()
        # the image IS the quantisation of the stored bf16 input at the slot's scale
        want8 = E.quantize_bits(xin.float().cpu() * (torch.tensor(1.0) / torch.tensor(sx, dtype=torch.float32)), E.E4M3)
        assert torch.equal(x8, want8), conv
        xd = E.dequantize_bits(x8, E.E4M3).double().permute(0, 3, 1, 2)
        wd = E.dequantize_bits(pk["wf8"].cpu(), E.E4M3).double()                # [9][Cout][Cin]
        cout, cin = wd.shape[1], wd.shape[2]
        w4 = wd.view(3, 3, cout, cin).permute(2, 3, 0, 1)
        ref = F.conv2d(xd, w4, None, padding=1) * (sx * sw) + eng.P[conv + ".bias"].double().cpu().view(1, -1, 1, 1)
        if relu:
            ref = F.relu(ref)
        got = y.float().cpu().permute(0, 3, 1, 2).double()
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err <= 4e-3, (conv, err)          # one bf16 ulp of the stored output

