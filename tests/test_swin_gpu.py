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
@pytest.mark.parametrize("C_,M", [(96, 70), (384, 33), (1536, 9), (768, 130)])
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
@pytest.mark.parametrize("H,Wd,heads,shift", [(14, 14, 3, 0), (16, 16, 3, 3), (9, 12, 6, 3), (7, 7, 12, 0)])
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
