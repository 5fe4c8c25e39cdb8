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
    assert st[0].item() == amax                       # this step's amax (the recorded launch re-took it)
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
    assert slots.t[i, 0].item() == x2[..., 16:16 + C].float().abs().max().item()


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
