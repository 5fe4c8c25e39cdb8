"""Per-kernel parity tests (GPU): every C-ABI entry point against the same ATen CPU op the reference's modules
issue (fp32 / fp64 on the host).  Tolerances: f32 path 1e-4 relative to the tensor's magnitude (exact-f32 MFMA,
different summation order); bf16 path 2e-2 (8-bit mantissa in/out, f32 accumulate)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cvcs_amd import ops  # noqa: E402

DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16]


def tol(dtype):
    return 1e-4 if dtype == torch.float32 else 2e-2


def to_nhwc(x, dtype, cpad=None):
    """CPU NCHW f32 -> device NHWC `dtype` (optionally zero-padded channels)."""
    x = x.permute(0, 2, 3, 1).contiguous()
    if cpad is not None and cpad > x.shape[3]:
        x = F.pad(x, (0, cpad - x.shape[3]))
    return x.to(dtype).to(DEV).contiguous()


def from_nhwc(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def rq(x, dtype):
    """round a CPU f32 tensor through `dtype` (what the device operand actually holds)."""
    return x.to(dtype).float()


def close(got, ref, rel, what=""):
    scale = max(1e-6, ref.abs().max().item())
    err = (got - ref).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.2e} > {rel})"


CONV_CASES = [
    # B, H, W, Cin, Cout, K, stride, pad, dil, relu
    (2, 9, 7, 32, 64, 3, 1, 1, 1, False),      # ragged M (126 pixels), one partial tile
    (1, 16, 16, 64, 128, 3, 1, 1, 1, True),    # BN=128 path, ReLU epilogue
    (3, 12, 20, 32, 192, 3, 1, 1, 1, False),   # Cout % 128 != 0 -> BN=64, M=720 (5.6 tiles)
    (2, 8, 8, 64, 64, 1, 1, 0, 1, False),      # 1x1
    (2, 16, 16, 64, 64, 2, 2, 0, 1, False),    # 2x2 stride 2 (ConvTranspose data-gradient form)
    (1, 20, 20, 32, 64, 3, 1, 2, 2, False),    # dilation 2 (ASPP-style)
    (1, 17, 13, 32, 64, 3, 2, 1, 1, True),     # stride 2, odd extent
    (2, 2, 2, 128, 256, 3, 1, 1, 1, False),    # deepest level of a 32x32 tile: 2x2 maps
    (2, 32, 48, 64, 128, 3, 1, 1, 1, True),    # halo kernel, whole 16x16 tiles, BN=128
    (1, 28, 20, 128, 64, 3, 1, 1, 1, False),   # halo kernel, ragged tiles (28x20), BN=64, 4 channel slices
    (3, 14, 14, 96, 192, 3, 1, 1, 1, True),    # halo kernel, 14x14 maps (224-tile level 5), Cout % 128 != 0
    (1, 40, 24, 64, 256, 3, 1, 1, 1, False),   # halo kernel, two column tiles of weights
    (2, 32, 32, 128, 32, 3, 1, 1, 1, False),   # thin output (light decoder stage 128 -> 32): masked 64-channel tile of the halo kernel
    (1, 24, 40, 32, 16, 3, 1, 1, 1, True),     # thin 32 -> 16, ragged tiles
    (2, 16, 16, 64, 48, 3, 1, 1, 1, False),    # 48 channels (the DeepLabV3+ paper's low-level width)
    (2, 32, 64, 16, 16, 3, 1, 1, 1, False),    # thin kernel 16 -> 16, whole 4 x 64 tiles
    (1, 21, 100, 16, 32, 3, 1, 1, 1, True),    # thin 16 -> 32, ragged tiles both ways, ReLU
    (2, 18, 40, 32, 32, 3, 1, 1, 1, False),    # thin 32 -> 32
    (4, 128, 128, 32, 16, 3, 1, 1, 1, False),  # thin, > 1024 partial-statistics rows: the two-stage BatchNorm merge with C = 16
    (4, 128, 128, 128, 32, 3, 1, 1, 1, False), # masked halo tile, two-stage merge with C = 32
    (2, 16, 24, 256, 512, 1, 2, 0, 1, False),  # 1x1 / stride 2 (projection shortcut of a down-sampling block): the taps kernel reads the even pixels in place
    (3, 8, 8, 64, 128, 1, 2, 0, 1, False),     # 1x1 / stride 2, one K-slice pair, ragged last pixel tile
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_matches_aten(case, dtype):
    """S/blocks.py:3-4 nn.Conv2d; epilogue bias/ReLU and the fused BatchNorm statistics."""
    B, H, W, Cin, Cout, K, stride, pad, dil, relu = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = rq(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = rq(torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5, dtype)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad, dilation=dil)
    if relu:
        ref = ref.relu()
    ref = ref.float()
    xd = to_nhwc(x, dtype)
    wf, _ = ops.pack_conv_weight(w.to(DEV), max(Cin, ops.KGROUP[ops.dtype_code(dtype)]), dtype, want_dgrad=False)
    Ho, Wo = ops.conv_out_hw(H, W, K, K, stride, pad, dil)
    out = torch.full((B, Ho, Wo, Cout + 64), 7.0, dtype=dtype, device=DEV)  # wider buffer: checks ld / channel offset
    M = B * Ho * Wo
    rows = ops.conv_stat_rows(ops.view(xd), Cout, K, K, stride, pad, dil)
    stats = (torch.zeros(rows, Cout, device=DEV), torch.zeros(rows, Cout, device=DEV), torch.zeros(rows, device=DEV))
    ops.conv2d(ops.view(xd), wf, b.to(DEV), ops.View(out, 64, Cout), K, K, stride, pad, dil, relu=relu, stats=stats)
    torch.cuda.synchronize()
    got = from_nhwc(out[..., 64:])
    close(got, ref, tol(dtype), "conv out")
    assert (out[..., :64].float() == 7.0).all(), "kernel wrote outside its channel range"
    # fused statistics against the f64 statistics of the exact result
    gamma, beta = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
    rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    scale, shift, mean, invstd = (torch.empty(Cout, device=DEV) for _ in range(4))
    ops.bn_finalize(stats, rows, M, Cout, gamma, beta, rm, rv, True, scale, shift, mean, invstd)
    torch.cuda.synchronize()
    assert stats[2].sum().item() == M
    rmean = ref.double().mean(dim=(0, 2, 3))
    rvar = ref.double().var(dim=(0, 2, 3), unbiased=False)
    close(mean.cpu().double(), rmean, 5e-3 if dtype == torch.bfloat16 else 1e-4, "batch mean")
    close(invstd.cpu().double(), 1.0 / torch.sqrt(rvar + 1e-5), 5e-3 if dtype == torch.bfloat16 else 1e-4, "batch invstd")
    unb = ref.double().var(dim=(0, 2, 3), unbiased=True) if M > 1 else rvar
    close(rv.cpu().double(), 0.9 + 0.1 * unb, 5e-3 if dtype == torch.bfloat16 else 1e-4, "running var")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,W", [(24, 40), (6, 6)])
def test_conv2d_narrow_first_layer_input(H, W, dtype):
    """the first layer (S/nets.py:126, 3 -> 64): the tile is stored 16 bytes per pixel (8 bf16 / 4 f32 channels) and the
    contraction runs over one zero-padded K-group whose padding is never fetched (cvcs_conv_desc.Cin_valid); halo kernel
    (24x40) and generic kernel (6x6).  The buffer behind the 16 bytes is poisoned: nothing beyond Cin_valid may be read."""
    g = torch.Generator().manual_seed(H)
    B, Cout = 2, 64
    x = torch.randint(0, 256, (B, 3, H, W), generator=g).float()          # raw 0..255 pixels: exact in bf16
    w = rq(torch.randn(Cout, 3, 3, 3, generator=g) / 27 ** 0.5, dtype)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).float()
    cs = 16 // torch.empty((), dtype=dtype).element_size()                  # stored channels: 8 | 4
    buf = torch.full((B, H, W, cs + 32), float("nan"), dtype=dtype, device=DEV)
    xin = ops.View(buf, 0, cs)
    packed = torch.empty((B, H, W, cs), dtype=dtype, device=DEV)
    ops.pack_input(x.to(DEV).contiguous(), packed)
    buf[..., :cs] = packed
    wf, _ = ops.pack_conv_weight(w.to(DEV), ops.KGROUP[ops.dtype_code(dtype)], dtype, want_dgrad=False)
    out = torch.empty(B, H, W, Cout, dtype=dtype, device=DEV)
    ops.conv2d(xin, wf, b.to(DEV), ops.view(out), 3, 3, 1, 1)
    torch.cuda.synchronize()
    close(from_nhwc(out), ref, tol(dtype), "narrow-input conv")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("order", ["encoder", "decoder"])
@pytest.mark.parametrize("H,W,Cin,Cout,pool", [(20, 36, 64, 128, True), (6, 6, 64, 64, False), (16, 16, 128, 64, True)])
def test_conv2d_eval_mode_batchnorm_fold(H, W, Cin, Cout, pool, order, dtype):
    """eval mode: BN (running statistics) folded into the conv epilogue - encoder order conv->BN->ReLU (S/blocks.py:13-17)
    and decoder order conv->ReLU->BN (S/blocks.py:40-45) - plus the pooling-only pass (S/nets.py:130); halo kernel
    (20x36, 16x16) and generic kernel (6x6 map)."""
    g = torch.Generator().manual_seed(H * 7 + Cout)
    B = 2
    x = rq(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = rq(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5, dtype)
    b = torch.randn(Cout, generator=g)
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    rm, rv = torch.randn(Cout, generator=g), torch.rand(Cout, generator=g) + 0.5
    y = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    bn = lambda t: F.batch_norm(t, rm.double(), rv.double(), gamma.double(), beta.double(), False, 0.1, 1e-5)
    ref = (bn(y).relu() if order == "encoder" else bn(y.relu())).float()
    scale, shift = torch.empty(Cout, device=DEV), torch.empty(Cout, device=DEV)
    ops.bn_finalize(None, 0, B * H * W, Cout, gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV), False, scale, shift, None, None)
    wf, _ = ops.pack_conv_weight(w.to(DEV), Cin, dtype, want_dgrad=False)
    out = torch.empty(B, H, W, Cout, dtype=dtype, device=DEV)
    fold = (scale, shift)
    fused = torch.full((B, H // 2, W // 2, Cout + 64), 7.0, dtype=dtype, device=DEV) if pool and H >= 8 else None
    ops.conv2d(ops.view(to_nhwc(x, dtype)), wf, b.to(DEV), ops.view(out), 3, 3, 1, 1, relu=True,
               pre_affine=fold if order == "encoder" else None, post_affine=fold if order == "decoder" else None,
               pool=None if fused is None else ops.View(fused, 64, Cout))
    torch.cuda.synchronize()
    close(from_nhwc(out), ref, tol(dtype), "folded conv+BN")
    if fused is not None:     # MaxPool2d(2,2) fused into the epilogue of the halo kernel: exactly the pooled stored values
        assert torch.equal(from_nhwc(fused[..., 64:]), F.max_pool2d(from_nhwc(out), 2, 2))
        assert (fused[..., :64].float() == 7.0).all()
    if pool:
        pl = torch.empty(B, H // 2, W // 2, Cout, dtype=dtype, device=DEV)
        keep = out.clone()
        ops.bn_act(ops.view(out), torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV), False, None, ops.view(pl))
        torch.cuda.synchronize()
        assert torch.equal(out, keep)                                   # pooling-only: the source is not rewritten
        assert torch.equal(from_nhwc(pl), F.max_pool2d(from_nhwc(out), 2, 2))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W", [(2, 16, 16), (4, 192, 192)])  # the second has 2304 partial rows: two-stage merge
def test_conv2d_f32_large_mean_statistics(dtype, B, H, W):
    """raw 0..255 inputs (S/train.py:121, no normalisation): mean >> std must not cancel in the variance."""
    g = torch.Generator().manual_seed(5)
    Cin, Cout = ops.KGROUP[ops.dtype_code(dtype)], 64
    x = torch.zeros(B, Cin, H, W)
    x[:, :3] = torch.randint(0, 256, (B, 3, H, W), generator=g).float()
    w = rq(torch.rand(Cout, Cin, 3, 3, generator=g) * 0.1, dtype)  # all-positive weights: large positive mean
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    wf, _ = ops.pack_conv_weight(w.to(DEV), Cin, dtype, want_dgrad=False)
    out = torch.empty(B, H, W, Cout, dtype=dtype, device=DEV)
    M = B * H * W
    xv = ops.view(to_nhwc(x, dtype))
    rows = ops.conv_stat_rows(xv, Cout, 3, 3, 1, 1)
    stats = (torch.zeros(rows, Cout, device=DEV), torch.zeros(rows, Cout, device=DEV), torch.zeros(rows, device=DEV))
    ops.conv2d(xv, wf, None, ops.view(out), 3, 3, 1, 1, stats=stats)
    gamma, beta = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
    rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    scale, shift, mean, invstd = (torch.empty(Cout, device=DEV) for _ in range(4))
    ops.bn_finalize(stats, rows, M, Cout, gamma, beta, rm, rv, True, scale, shift, mean, invstd)
    torch.cuda.synchronize()
    rvar = ref.var(dim=(0, 2, 3), unbiased=False)
    close(invstd.cpu().double(), 1.0 / torch.sqrt(rvar + 1e-5), 1e-2 if dtype == torch.bfloat16 else 1e-4, "invstd")
    close(mean.cpu().double(), ref.mean(dim=(0, 2, 3)), 1e-2 if dtype == torch.bfloat16 else 1e-5, "mean")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 4, 4, 128, 64), (1, 7, 5, 256, 128), (2, 20, 24, 128, 64), (1, 32, 32, 512, 256)])
def test_conv_transpose_forward_and_dgrad(shape, dtype):
    """nn.ConvTranspose2d(k2,s2) (S/nets.py:150,156,162,168) = 1x1 conv + pixel shuffle; its data gradient =
    2x2 stride-2 conv."""
    B, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(3)
    x = rq(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = rq(torch.randn(Cin, Cout, 2, 2, generator=g) / Cin ** 0.5, dtype)
    b = torch.randn(Cout, generator=g)
    ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2).float()
    wf = torch.empty(1, 4 * Cout, Cin, dtype=dtype, device=DEV)
    wd = torch.empty(4, Cin, Cout, dtype=dtype, device=DEV)
    b4 = torch.empty(4 * Cout, device=DEV)
    ops.pack_convT_weight_into(w.to(DEV), b.to(DEV), wf, b4, wd)
    out = torch.zeros(B, 2 * H, 2 * W, 2 * Cout, dtype=dtype, device=DEV)  # second half of a concat buffer
    ops.conv2d(ops.view(to_nhwc(x, dtype)), wf, b4, ops.View(out, Cout, Cout), 1, 1, pixel_shuffle=True)
    torch.cuda.synchronize()
    close(from_nhwc(out[..., Cout:]), ref, tol(dtype), "convT fwd")
    assert (out[..., :Cout] == 0).all()
    # data gradient
    gy = rq(torch.randn(B, Cout, 2 * H, 2 * W, generator=g), dtype)
    xr = x.double().requires_grad_(True)
    F.conv_transpose2d(xr, w.double(), None, stride=2).backward(gy.double())
    gin = torch.empty(B, H, W, Cin, dtype=dtype, device=DEV)
    ops.conv2d(ops.view(to_nhwc(gy, dtype)), wd, None, ops.view(gin), 2, 2, stride=2, pad=0)
    torch.cuda.synchronize()
    close(from_nhwc(gin), xr.grad.float(), tol(dtype), "convT dgrad")


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_dgrad_through_flipped_weights(dtype):
    """input-gradient half of convolution_backward (S/train.py:125) for a 3x3 / pad 1 conv."""
    g = torch.Generator().manual_seed(9)
    B, H, W, Cin, Cout = 2, 10, 6, 64, 128
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = rq(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5, dtype)
    gy = rq(torch.randn(B, Cout, H, W, generator=g), dtype)
    F.conv2d(x, w.double(), None, padding=1).backward(gy.double())
    _, wd = ops.pack_conv_weight(w.to(DEV), Cin, dtype)
    gin = torch.empty(B, H, W, Cin, dtype=dtype, device=DEV)
    ops.conv2d(ops.view(to_nhwc(gy, dtype)), wd, None, ops.view(gin), 3, 3, 1, 1)
    torch.cuda.synchronize()
    close(from_nhwc(gin), x.grad.float(), tol(dtype), "dgrad")


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("shape", [(2, 32, 48, 64, 64), (1, 24, 40, 128, 128), (2, 20, 36, 128, 64), (1, 16, 16, 64, 256),
                                   (2, 32, 64, 16, 16), (1, 21, 100, 32, 32), (3, 18, 40, 16, 32), (2, 64, 128, 32, 16)])
def test_conv_launch_takes_the_first_pass_of_the_batchnorm_backward(shape, mode):
    """cvcs_conv_desc.bwd_y: the data-gradient launch above a conv->BN->ReLU (mode 0) / conv->ReLU->BN (mode 1) block
    (S/blocks.py:13-17, :40-45) also sums dz and dz * xhat of that block over the gradient it stores - compared with
    cvcs_bn_bwd_reduce on the stored output, and the output itself with the plain launch (bit for bit)."""
    B, H, W, Cin, Cout = shape
    thin = Cout <= 32       # the thin kernel of the light decoder stages: mode 0 only, one partial row per workgroup
    if thin and mode == 1:
        pytest.skip("the thin kernel carries the reduce of conv -> BN -> ReLU blocks only")
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(31 + mode)
    gy = rq(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = rq(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5, dtype)
    wf, _ = ops.pack_conv_weight(w.to(DEV), max(Cin, 32), dtype)      # (a 16-channel input runs under a zero-padded 32-channel K-group)
    xin = ops.view(to_nhwc(gy, dtype))
    y = torch.randn(B, H, W, Cout, generator=g)
    if mode == 1:
        y = y.relu()
    y = y.to(dtype).to(DEV)
    scale, shift, mean, invstd = (torch.randn(Cout, generator=g).to(DEV) for _ in range(4))
    invstd = invstd.abs() + 0.5
    plain = torch.empty(B, H, W, Cout, dtype=dtype, device=DEV)
    ops.conv2d(xin, wf, None, ops.view(plain), 3, 3, 1, 1)
    rows = ops.bn_bwd_rows(B * H * W)
    r0, r1 = torch.zeros(rows * Cout, device=DEV), torch.zeros(rows * Cout, device=DEV)
    ops.bn_bwd_reduce(ops.view(y), ops.view(plain), None, scale, shift, mean, invstd, mode, r0, r1)
    trows = ops.conv_stat_rows(xin, Cout, 3, 3, 1, 1, bwd=True)
    assert (1 <= trows <= B * ((H + 3) // 4) * ((W + 63) // 64)) if thin else trows == B * ((H + 15) // 16) * ((W + 15) // 16)
    p0, p1 = torch.full((trows * Cout,), float("nan"), device=DEV), torch.full((trows * Cout,), float("nan"), device=DEV)
    fused = torch.empty_like(plain)
    ops.conv2d(xin, wf, None, ops.view(fused), 3, 3, 1, 1, bn_bwd=(ops.view(y), scale, shift, mean, invstd, mode, p0, p1))
    torch.cuda.synchronize()
    assert torch.equal(fused, plain)
    for got, ref, what in ((p0, r0, "sum dz"), (p1, r1, "sum dz*xhat")):
        a = got.view(trows, Cout).double().sum(0).cpu()
        b = ref.view(rows, Cout).double().sum(0).cpu()
        assert torch.isfinite(a).all()
        close(a, b, 1e-4, what)


@pytest.mark.parametrize("B,Hl,Wl", [(2, 8, 32), (1, 10, 24), (2, 6, 40), (1, 64, 64)])
def test_thin_conv_and_weight_gradient_over_an_upsampled_input_equal_the_stored_route(B, Hl, Wl):
    """cvcs_conv_desc.in_up2 / cvcs_wgrad_desc.x_up2 (opt-in in the engine: CVCS_FUSE_UPSAMPLE): nn.Upsample(2, 'bilinear') -> Conv2d(32, 16, 3, padding=1)
    of the light decoder's full-resolution stage with the up-sampling interpolated while the thin kernels stage their input - conv output,
    BatchNorm partial statistics and weight gradient BIT-IDENTICAL to cvcs_upsample2x_fwd followed by the same launches on the stored tensor
    (whole and ragged 4 x 64 tiles, borders of the replicate-clamped bilinear kernel and of the conv's zero padding)"""
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(B * Hl + Wl)
    Cin, Cout, H, W = 32, 16, 2 * Hl, 2 * Wl
    xl = torch.randn(B, Hl, Wl, Cin, generator=g).to(dtype).to(DEV)
    up = torch.empty(B, H, W, Cin, dtype=dtype, device=DEV)
    ops.upsample2x_fwd(ops.view(xl), ops.view(up))
    w = rq(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5, dtype)
    wf, _ = ops.pack_conv_weight(w.to(DEV), Cin, dtype)
    outs, stats = [], []
    for xin in (ops.view(up), ops.up_view(ops.view(xl))):
        assert (xin.H, xin.W) == (H, W)
        rows = ops.conv_stat_rows(xin, Cout, 3, 3, 1, 1)
        st = (torch.zeros(rows, Cout, device=DEV), torch.zeros(rows, Cout, device=DEV), torch.zeros(rows, device=DEV))
        out = torch.full((B, H, W, Cout), 7.0, dtype=dtype, device=DEV)
        ops.conv2d(xin, wf, None, ops.view(out), 3, 3, 1, 1, stats=st)
        outs.append(out); stats.append(st)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    assert all(torch.equal(a, b) for a, b in zip(stats[0], stats[1]))
    ref = F.conv2d(F.interpolate(from_nhwc(xl), scale_factor=2, mode="bilinear", align_corners=False).to(dtype).float(), w, None, padding=1)
    close(from_nhwc(outs[1]), ref, 2.0 ** -7, "conv over the up-sampled input")
    dy = torch.randn(B, H, W, Cout, generator=g).to(dtype).to(DEV)
    dws = []
    for xin in (ops.view(up), ops.up_view(ops.view(xl))):
        dw = torch.full((Cout, Cin, 3, 3), float("nan"), device=DEV)
        ws = torch.empty(ops.wgrad_workspace_floats_for(xin, ops.view(dy), 3, 3, 1, 1), device=DEV)
        ops.conv2d_wgrad(xin, ops.view(dy), dw, 3, 3, 1, 1, ws)
        dws.append(dw)
    torch.cuda.synchronize()
    assert torch.isfinite(dws[1]).all() and torch.equal(dws[0], dws[1])


WGRAD_CASES = [
    # B, H, W, Cin(stored), Cin_real, Cout, K, stride, pad
    (2, 8, 40, 64, 64, 64, 3, 1, 1),     # TW=32 strips, ragged width
    (3, 6, 6, 128, 128, 64, 3, 1, 1),    # TW=8, TH=4
    (2, 2, 2, 64, 64, 128, 3, 1, 1),     # 2x2 maps (deepest level of a 32x32 tile)
    (2, 16, 16, None, 3, 64, 3, 1, 1),   # first layer: 3 real channels zero-padded to one K-group
    (2, 8, 8, 64, 64, 128, 2, 2, 0),     # ConvTranspose weight gradient form (2x2, stride 2)
    (1, 14, 14, 64, 64, 64, 3, 1, 1),    # 14x14 (level 5 of a 224 tile)
    (2, 5, 9, 64, 64, 64, 1, 1, 0),      # 1x1
    (2, 7, 64, 128, 128, 64, 3, 1, 1),   # fast path: 1x32 K-tiles, two Cin tiles, whole rows
    (3, 5, 50, 64, 64, 128, 3, 1, 1),    # fast path: ragged row tiles (50 = 32 + 18)
    (1, 33, 96, 192, 192, 64, 3, 1, 1),  # fast path: 3 Cin tiles, many K-tiles per slice
    (2, 21, 44, 128, 128, 256, 3, 1, 1), # 128-channel fast path (8 waves): two Cout tiles x two Cin tiles, ragged 4x8 K-tiles
    (1, 40, 72, 72, 72, 128, 3, 1, 1),   # 128-channel fast path: Cin not a multiple of 64 (absent channels zero-filled)
    (2, 16, 32, 64, 64, 128, 2, 2, 0),   # stride-2 fast path (ConvTranspose weight gradient): whole 4x8 K-tiles
    (1, 24, 20, 128, 128, 256, 2, 2, 0), # stride-2 fast path: ragged K-tiles (Wo = 10), two Cout x two Cin tiles
    (3, 16, 16, 72, 72, 128, 2, 2, 0),   # stride-2 fast path: Cin not a multiple of 64
    (2, 32, 64, 16, 16, 16, 3, 1, 1),    # thin kernel (light decoder stages): 16 -> 16, whole 4 x 64 tiles
    (1, 21, 100, 32, 32, 16, 3, 1, 1),   # thin 32 -> 16, ragged tiles in both directions
    (2, 16, 72, 32, 32, 32, 3, 1, 1),    # thin 32 -> 32
    (1, 12, 40, 16, 16, 32, 3, 1, 1),    # thin 16 -> 32
    (2, 8, 24, 128, 128, 32, 3, 1, 1),   # Cout = 32 on the generic kernel (masked row tile, skipped zero blocks), narrow map
    (1, 16, 16, 64, 64, 16, 3, 1, 1),    # Cout = 16 on the generic kernel
    (2, 20, 72, 128, 128, 32, 3, 1, 1),  # thin Cout under a wide input (the 128 -> 32 decoder layer): four 32-channel groups of the thin kernel
    (1, 9, 40, 96, 96, 16, 3, 1, 1),     # thin 96 -> 16: three groups, ragged tiles
    (2, 16, 24, 256, 256, 512, 1, 2, 0), # 1x1 / stride 2 on the GEMM kernel: x read at the even pixels in place (no gathered copy)
    (3, 8, 12, 64, 64, 128, 1, 2, 0),    # 1x1 / stride 2, ragged last K-tile
    (2, 8, 8, 64, 64, 64, 1, 1, 0),      # 64 -> 64 1x1 on the 64 x 64 GEMM instance
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad_matches_autograd(case, dtype):
    B, H, W, Cin, Cin_real, Cout, K, stride, pad = case
    if Cin is None:
        Cin = ops.KGROUP[ops.dtype_code(dtype)]
    g = torch.Generator().manual_seed(21)
    x = rq(torch.randn(B, Cin_real, H, W, generator=g), dtype)
    Ho, Wo = ops.conv_out_hw(H, W, K, K, stride, pad)
    gy = rq(torch.randn(B, Cout, Ho, Wo, generator=g), dtype)
    w = torch.zeros(Cout, Cin_real, K, K, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, stride=stride, padding=pad).backward(gy.double())
    xv, gv = ops.view(to_nhwc(x, dtype, Cin)), ops.view(to_nhwc(gy, dtype))
    ws = torch.empty(max(ops.wgrad_workspace_floats(B, Ho, Wo, Cout, Cin, K, K, stride), ops.wgrad_workspace_floats_for(xv, gv, K, K, stride, pad)), device=DEV)
    dw = torch.empty(Cout, Cin_real, K, K, device=DEV)
    ops.conv2d_wgrad(xv, gv, dw, K, K, stride, pad, ws, cin_real=Cin_real)
    torch.cuda.synchronize()
    close(dw.cpu(), w.grad.float(), 1e-4 if dtype == torch.float32 else 1e-3, "dW")  # exact products, f32 sums


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("relu,pool", [(True, False), (True, True), (False, False)])
def test_bn_act_and_pool(relu, pool, dtype):
    """BN apply (+ReLU) (+MaxPool2d(2,2)): S/blocks.py:14-17, S/nets.py:130."""
    g = torch.Generator().manual_seed(2)
    B, H, W, C = 2, 6, 10, 128
    y = rq(torch.randn(B, C, H, W, generator=g) * 3, dtype)
    scale, shift = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = y * scale[None, :, None, None] + shift[None, :, None, None]
    if relu:
        ref = ref.relu()
    out = torch.zeros(B, H, W, 2 * C, dtype=dtype, device=DEV)
    pl = torch.empty(B, H // 2, W // 2, C, dtype=dtype, device=DEV) if pool else None
    ops.bn_act(ops.view(to_nhwc(y, dtype)), scale.to(DEV), shift.to(DEV), relu, ops.View(out, C, C),
               ops.view(pl) if pool else None)
    torch.cuda.synchronize()
    close(from_nhwc(out[..., C:]), ref, 1e-6 if dtype == torch.float32 else 8e-3, "bn_act")
    if pool:
        close(from_nhwc(pl), F.max_pool2d(ref, 2, 2), 1e-6 if dtype == torch.float32 else 8e-3, "pool")


def _bn_bwd_device(dtype, y, g1, g2, gamma, beta, mode):
    """run finalize(train) from exact stats + the two backward passes; returns dy, dgamma, dbeta, db."""
    B, C, H, W = y.shape
    M = B * H * W
    yd = to_nhwc(y, dtype)
    mean = y.double().mean(dim=(0, 2, 3))
    var = y.double().var(dim=(0, 2, 3), unbiased=False)
    invstd = (1.0 / torch.sqrt(var + 1e-5))
    scale = (gamma.double() * invstd).float().to(DEV)
    shift = (beta.double() - mean * gamma.double() * invstd).float().to(DEV)
    mean_d, invstd_d = mean.float().to(DEV), invstd.float().to(DEV)
    rows = ops.bn_bwd_rows(M)
    p0, p1, pb = (torch.zeros(rows, C, device=DEV) for _ in range(3))
    g1v = ops.view(to_nhwc(g1, dtype))
    g2v = ops.view(to_nhwc(g2, dtype)) if g2 is not None else None
    ops.bn_bwd_reduce(ops.view(yd), g1v, g2v, scale, shift, mean_d, invstd_d, mode, p0, p1)
    dgamma, dbeta, ca, cb, db = (torch.empty(C, device=DEV) for _ in range(5))
    ops.bn_bwd_finalize(p0, p1, rows, M, C, gamma.to(DEV), invstd_d, dgamma, dbeta, ca, cb)
    dy = torch.empty(B, H, W, C, dtype=dtype, device=DEV)
    ops.bn_bwd_apply(ops.view(yd), g1v, g2v, scale, shift, mean_d, invstd_d, ca, cb, mode, ops.view(dy), pb)
    ops.colsum_finalize(pb, rows, C, db)
    torch.cuda.synchronize()
    return from_nhwc(dy), dgamma.cpu(), dbeta.cpu(), db.cpu()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("pool", [False, True])
def test_bn_backward_encoder(pool, dtype):
    """backward of ReLU(BN(y)) [+ MaxPool2d] (UnetEncodeLayer, S/blocks.py:8-23; S/nets.py:130)."""
    g = torch.Generator().manual_seed(4)
    B, C, H, W = 2, 64, 8, 12
    y = rq(torch.randn(B, C, H, W, generator=g) * 2 + 0.5, dtype)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    g1 = rq(torch.randn(B, C, H, W, generator=g), dtype)
    g2 = rq(torch.randn(B, C, H // 2, W // 2, generator=g), dtype) if pool else None
    yr = y.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    a = F.relu(F.batch_norm(yr, None, None, gr, br, training=True, eps=1e-5))
    loss = (a * g1.double()).sum()
    if pool:
        loss = loss + (F.max_pool2d(a, 2, 2) * g2.double()).sum()
    loss.backward()
    dy, dgamma, dbeta, db = _bn_bwd_device(dtype, y, g1, g2, gamma, beta, 0)
    t = 2e-4 if dtype == torch.float32 else 2e-2
    close(dy, yr.grad.float(), t, "dy")
    close(dgamma, gr.grad.float(), t, "dgamma")
    close(dbeta, br.grad.float(), t, "dbeta")
    assert db.abs().max() <= t * yr.grad.abs().sum(dim=(0, 2, 3)).max().float() + 1e-3  # analytically zero


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn_backward_decoder(dtype):
    """backward of BN(ReLU(conv)) (UnetForwardDecodeLayer order, S/blocks.py:40-45): r is the stored tensor."""
    g = torch.Generator().manual_seed(6)
    B, C, H, W = 2, 128, 6, 6
    pre = rq(torch.randn(B, C, H, W, generator=g), dtype)
    r = pre.relu()
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    g1 = rq(torch.randn(B, C, H, W, generator=g), dtype)
    pr = pre.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    z = F.batch_norm(F.relu(pr), None, None, gr, br, training=True, eps=1e-5)
    (z * g1.double()).sum().backward()
    dy, dgamma, dbeta, db = _bn_bwd_device(dtype, r, g1, None, gamma, beta, 1)
    t = 2e-4 if dtype == torch.float32 else 2e-2
    close(dy, pr.grad.float(), t, "dy")
    close(dgamma, gr.grad.float(), t, "dgamma")
    close(dbeta, br.grad.float(), t, "dbeta")
    close(db, pr.grad.sum(dim=(0, 2, 3)).float(), 5 * t, "db")


@pytest.mark.parametrize("dtype", DTYPES)
def test_bilinear_upsample(dtype):
    """nn.Upsample(scale_factor=2, mode='bilinear') (S/blocks.py:29) forward and backward."""
    g = torch.Generator().manual_seed(8)
    B, C, H, W = 2, 64, 5, 7
    x = rq(torch.randn(B, C, H, W, generator=g), dtype)
    xr = x.double().requires_grad_(True)
    ref = F.interpolate(xr, scale_factor=(2, 2), mode="bilinear")
    gy = rq(torch.randn(B, C, 2 * H, 2 * W, generator=g), dtype)
    ref.backward(gy.double())
    out = torch.empty(B, 2 * H, 2 * W, C, dtype=dtype, device=DEV)
    ops.upsample2x_fwd(ops.view(to_nhwc(x, dtype)), ops.view(out))
    gin = torch.empty(B, H, W, C, dtype=dtype, device=DEV)
    ops.upsample2x_bwd(ops.view(to_nhwc(gy, dtype)), ops.view(gin))
    torch.cuda.synchronize()
    t = 1e-6 if dtype == torch.float32 else 8e-3
    close(from_nhwc(out), ref.detach().float(), t, "upsample fwd")
    close(from_nhwc(gin), xr.grad.float(), t, "upsample bwd")


@pytest.mark.parametrize("src_dtype", [torch.uint8, torch.float32])
@pytest.mark.parametrize("dtype", DTYPES)
def test_pack_input(dtype, src_dtype):
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 256, (2, 3, 6, 10), generator=g).to(src_dtype)
    cpad = ops.KGROUP[ops.dtype_code(dtype)]
    dst = torch.full((2, 6, 10, cpad), 5.0, dtype=dtype, device=DEV)
    ops.pack_input(x.to(DEV), dst)
    torch.cuda.synchronize()
    got = dst.float().cpu()
    assert torch.equal(got[..., :3], x.float().permute(0, 2, 3, 1))  # 0..255 is exact in bf16
    assert (got[..., 3:] == 0).all()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("NC,B,H,W", [(5, 2, 9, 11), (16, 2, 9, 11), (21, 2, 9, 11), (16, 3, 64, 96), (32, 1, 7, 5)])
def test_head_forward_backward(NC, B, H, W, dtype):
    """nn.Conv2d(64, NC, 1) (S/nets.py:172) forward (NHWC -> NCHW f32 logits) and backward (ragged last 64-pixel tile,
    more tiles than workgroup rows, the 32-class limit)."""
    g = torch.Generator().manual_seed(12)
    C = 64
    x = rq(torch.randn(B, C, H, W, generator=g), dtype)
    w = torch.randn(NC, C, generator=g) / 8
    b = torch.randn(NC, generator=g)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv2d(xr, wr[:, :, None, None], br)
    dl = torch.randn(B, NC, H, W, generator=g)
    ref.backward(dl.double())
    xd = to_nhwc(x, dtype)
    logits = torch.empty(B, NC, H, W, device=DEV)
    ops.head_fwd(ops.view(xd), w.to(DEV), b.to(DEV), logits)
    dx = torch.empty(B, H, W, C, dtype=dtype, device=DEV)
    rows = ops.head_bwd_rows(B * H * W)
    part = torch.zeros(rows, NC * C + NC, device=DEV)
    ops.head_bwd(ops.view(xd), dl.to(DEV), w.to(DEV), ops.view(dx), part)
    dwb = torch.empty(NC * C + NC, device=DEV)
    ops.colsum_finalize(part, rows, NC * C + NC, dwb)
    torch.cuda.synchronize()
    close(logits.cpu(), ref.detach().float(), 1e-5, "logits")
    close(from_nhwc(dx), xr.grad.float(), 1e-5 if dtype == torch.float32 else 8e-3, "head dx")
    close(dwb[:NC * C].cpu().reshape(NC, C), wr.grad.float(), 1e-4, "head dW")
    close(dwb[NC * C:].cpu(), br.grad.float(), 1e-4, "head db")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,NC,B,H,W", [(16, 16, 2, 24, 40), (32, 5, 1, 17, 13), (16, 21, 3, 32, 32), (32, 32, 1, 40, 40)])
def test_thin_head_forward_argmax_backward(C, NC, B, H, W, dtype):
    """the head on 16 / 32 input channels (decoder widths 256/128/64/32/16): logits, fused argmax = argmax of the logits, dx, dW, db;
    ragged last 256-pixel tile and more tiles than workgroup rows"""
    g = torch.Generator().manual_seed(C + NC)
    x = rq(torch.randn(B, C, H, W, generator=g), dtype)
    w = torch.randn(NC, C, generator=g) / 4
    b = torch.randn(NC, generator=g)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv2d(xr, wr[:, :, None, None], br)
    dl = torch.randn(B, NC, H, W, generator=g)
    ref.backward(dl.double())
    xd = torch.full((B, H, W, C + 16), 9.0, dtype=dtype, device=DEV)       # a view with a leading dimension
    xd[..., :C] = to_nhwc(x, dtype)
    xv = ops.View(xd, 0, C)
    logits = torch.empty(B, NC, H, W, device=DEV)
    ops.head_fwd(xv, w.to(DEV), b.to(DEV), logits)
    labels = torch.empty(B, H, W, dtype=torch.uint8, device=DEV)
    ops.head_argmax(xv, w.to(DEV), b.to(DEV), labels)
    dx = torch.full((B, H, W, C + 16), 3.0, dtype=dtype, device=DEV)
    rows = ops.head_bwd_rows(B * H * W)
    part = torch.zeros(rows, NC * C + NC, device=DEV)
    ops.head_bwd(xv, dl.to(DEV), w.to(DEV), ops.View(dx, 0, C), part)
    dwb = torch.empty(NC * C + NC, device=DEV)
    ops.colsum_finalize(part, rows, NC * C + NC, dwb)
    torch.cuda.synchronize()
    close(logits.cpu(), ref.detach().float(), 1e-5, "logits")
    assert torch.equal(labels.cpu().long(), logits.argmax(1).cpu())
    close(from_nhwc(dx[..., :C]), xr.grad.float(), 1e-5 if dtype == torch.float32 else 8e-3, "head dx")
    assert (dx[..., C:].float() == 3.0).all()
    close(dwb[:NC * C].cpu().reshape(NC, C), wr.grad.float(), 1e-4, "head dW")
    close(dwb[NC * C:].cpu(), br.grad.float(), 1e-4, "head db")


@pytest.mark.parametrize("tdtype", [torch.uint8, torch.int64])
@pytest.mark.parametrize("NC,ignore,weighted", [(5, 0, False), (16, 0, True), (16, -100, False), (21, -100, True)])
def test_cross_entropy(NC, ignore, weighted, tdtype):
    """nn.CrossEntropyLoss(weight, ignore_index) (S/utils.py:230,238) loss and d(loss)/d(logits)."""
    g = torch.Generator().manual_seed(NC)
    B, H, W = 3, 13, 17
    z = torch.randn(B, NC, H, W, generator=g) * 5
    t = torch.randint(0, NC, (B, H, W), generator=g)
    cw = (torch.rand(NC, generator=g) * 2 + 0.1) if weighted else None
    if weighted and ignore == 0:
        cw[0] = 0.0
    zr = z.clone().requires_grad_(True)
    ref = F.cross_entropy(zr, t, weight=cw, ignore_index=ignore)
    ref.backward()
    P = B * H * W
    loss = torch.zeros(1, device=DEV)
    dz = torch.empty(B, NC, H, W, device=DEV)
    ws = torch.zeros(ops.ce_workspace_floats(P), device=DEV)
    ops.ce_fwd_bwd(z.to(DEV), t.to(tdtype).to(DEV), None if cw is None else cw.to(DEV), ignore, 1.0, loss, dz, ws)
    torch.cuda.synchronize()
    assert abs(loss.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    close(dz.cpu(), zr.grad, 1e-5, "dlogits")


def test_cross_entropy_all_pixels_ignored():
    """every target == ignore_index: torch returns a NaN loss and a ZERO gradient (the reference's step S/train.py:121-126
    then leaves the weights finite) - so must the fused launch, also with an external (sharded) denominator of 0."""
    g = torch.Generator().manual_seed(1)
    B, NC, H, W = 2, 5, 9, 11
    z = torch.randn(B, NC, H, W, generator=g)
    t = torch.zeros(B, H, W, dtype=torch.int64)
    zr = z.clone().requires_grad_(True)
    ref = F.cross_entropy(zr, t, ignore_index=0)
    ref.backward()
    assert torch.isnan(ref) and zr.grad.abs().max().item() == 0.0
    P = B * H * W
    for ext in (False, True):
        loss = torch.zeros(1, device=DEV)
        dz = torch.full((B, NC, H, W), 7.0, device=DEV)
        ws = torch.zeros(ops.ce_workspace_floats(P), device=DEV)
        if ext:
            ops.ce_weight_sum(t.to(DEV), B, NC, None, 0, ws)
        ops.ce_fwd_bwd(z.to(DEV), t.to(DEV), None, 0, 1.0, loss, dz, ws, external_denominator=ext)
        torch.cuda.synchronize()
        assert torch.isnan(loss).all() or ext, loss        # 0/0 like torch (sharded: this rank's share of a global mean)
        assert dz.abs().max().item() == 0.0


def test_argmax_and_confusion():
    """torch.max(dim=0) first-max ties (S/utils.py:90) + MulticlassConfusionMatrix(16, ignore_index=0) (S/utils.py:76-78)."""
    from oracle import unet_oracle as O
    g = torch.Generator().manual_seed(0)
    B, NC, H, W = 2, 16, 24, 20
    z = torch.randint(-3, 4, (B, NC, H, W), generator=g).float()  # many exact ties
    t = torch.randint(0, NC, (B, H, W), generator=g).to(torch.uint8)
    lab = torch.empty(B, H, W, dtype=torch.uint8, device=DEV)
    conf = torch.zeros(16, 16, dtype=torch.int64, device=DEV)
    for _ in range(2):  # accumulates across calls
        ops.argmax_confusion(z.to(DEV), lab, t.to(DEV), 0, 16, conf)
    torch.cuda.synchronize()
    ref_lab = torch.argmax(z, dim=1)
    assert torch.equal(lab.cpu().long(), ref_lab)
    ref_conf = O.confusion_matrix(ref_lab.numpy(), t.numpy(), 16, ignore_index=0)
    assert (conf.cpu().numpy() == 2 * ref_conf).all()


def test_optimizers_match_torch():
    """torch.optim.SGD(0.006, 0.9, wd 1e-5) / torch.optim.Adam(0.005) (S/utils.py:214,217), 3 steps."""
    g = torch.Generator().manual_seed(0)
    n = 1003  # not a multiple of 4: tail path
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) for _ in range(3)]
    for kind in ("sgd", "adam"):
        pr = torch.nn.Parameter(p0.clone())
        opt = torch.optim.SGD([pr], lr=0.006, momentum=0.9, weight_decay=1e-5) if kind == "sgd" else \
            torch.optim.Adam([pr], lr=0.005)
        p = p0.clone().to(DEV)
        s1, s2 = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        for i, gr in enumerate(grads):
            pr.grad = gr.clone()
            opt.step()
            if kind == "sgd":
                ops.sgd_step(p, gr.to(DEV), s1, 0.006, 0.9, 1e-5, 1.0, i == 0)
            else:
                ops.adam_step(p, gr.to(DEV), s1, s2, 0.005, 0.9, 0.999, 1e-8, 0.0, 1.0, i + 1)
        torch.cuda.synchronize()
        np.testing.assert_allclose(p.cpu().numpy(), pr.detach().numpy(), rtol=2e-5, atol=2e-6, err_msg=kind)


@pytest.mark.parametrize("dtype", DTYPES)
def test_batched_weight_pack_equals_per_layer_pack(dtype):
    """cvcs_pack_conv_weights (one launch over a device table, LDS-tiled) == cvcs_pack_conv_weight per layer, bit for bit:
    forward operand [tap][Cout][Cin_pad] and flipped data-gradient operand [tap'][Cin][Cout]; first layer (3 real channels
    in one zero-padded K-group, no dgrad operand), channel counts that are not multiples of the 64-channel block."""
    g = torch.Generator().manual_seed(8)
    kg = ops.KGROUP[ops.dtype_code(dtype)]
    shapes = [(64, 3, kg, False), (64, 64, 64, True), (128, 192, 192, True), (256, 128, 128, True), (16, 40, 64, True)]
    entries, want = [], []
    for cout, cin, cpad, dgrad in shapes:
        w = torch.randn(cout, cin, 3, 3, generator=g).to(DEV)
        wf, wd = ops.pack_conv_weight(w, cpad, dtype, want_dgrad=dgrad)
        wf2 = torch.full_like(wf, 7.0)
        wd2 = torch.full_like(wd, 7.0) if dgrad else None
        entries.append((w, wf2, wd2)); want.append((wf, wd))
    table, n = ops.pack_table(entries, DEV)
    ops.pack_conv_weights(table, n, dtype)
    torch.cuda.synchronize()
    for (w, wf2, wd2), (wf, wd) in zip(entries, want):
        assert torch.equal(wf2, wf), tuple(w.shape)
        assert wd is None or torch.equal(wd2, wd), tuple(w.shape)
