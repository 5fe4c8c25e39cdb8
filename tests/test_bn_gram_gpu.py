"""BatchNorm behind a 1x1 convolution without a stored convolution output (csrc/bn_gram.hip, cvcs_gram, the residual epilogue and the
second contraction source of cvcs_conv2d) against float64 torch on the host: the tail of torchvision's Bottleneck (conv1x1 -> BatchNorm ->
+ shortcut -> ReLU; the encoders of S/nets.py:234-311) forward and backward, entry point by entry point and as one chain against autograd."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cvcs_amd import ops  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


def _rq(x):
    return x.to(BF).float()


def _dev_view(x_nhwc_f32, pad_left=0, pad_right=0):
    """f32 CPU [B,H,W,C] -> a channel-range view of a wider bf16 device tensor (exercises ld != C)"""
    B, H, W, C_ = x_nhwc_f32.shape
    t = torch.full((B, H, W, pad_left + C_ + pad_right), 7.0, dtype=BF, device=DEV)
    t[..., pad_left:pad_left + C_] = x_nhwc_f32.to(BF).to(DEV)
    return ops.View(t, pad_left, C_)


def close(got, ref, rel, what=""):
    scale = max(1e-30, ref.abs().max().item())
    err = (got.double() - ref.double()).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.2e} > {rel})"


GRAM_CASES = [(2, 16, 16, 64, 0), (1, 37, 29, 128, 64), (2, 24, 24, 256, 0), (1, 9, 7, 512, 32), (4, 64, 64, 64, 192)]


@pytest.mark.parametrize("case", GRAM_CASES)
def test_gram_matrix_and_column_sums(case):
    B, H, W, C_, pad = case
    g = torch.Generator().manual_seed(C_ + H)
    a = _rq(torch.randn(B, H, W, C_, generator=g).relu() + 0.25)
    av = _dev_view(a, pad, 8 if pad else 0)
    M = B * H * W
    G = torch.empty(C_, C_, dtype=torch.float64, device=DEV)
    s = torch.empty(C_, dtype=torch.float64, device=DEV)
    ws = torch.empty(ops.gram_workspace_floats(M, C_), dtype=torch.float32, device=DEV)
    ops.gram(av, G, s, ws)
    torch.cuda.synchronize()
    a2 = a.reshape(M, C_).double()
    close(G.cpu(), a2.T @ a2, 2e-6, "gram")
    close(s.cpu(), a2.sum(0), 2e-6, "column sums")
    G2, s2 = torch.empty_like(G), torch.empty_like(s)
    ops.gram(av, G2, s2, ws)
    assert torch.equal(G, G2) and torch.equal(s, s2), "gram is not bitwise reproducible"


def _problem(B, H, W, m, C_, seed, mean_shift=0.0):
    g = torch.Generator().manual_seed(seed)
    a = _rq(torch.randn(B, H, W, m, generator=g).relu() + mean_shift)
    w = _rq(torch.randn(C_, m, generator=g) / m ** 0.5)
    gamma = torch.rand(C_, generator=g) + 0.5
    beta = torch.randn(C_, generator=g) * 0.3
    return a, w, gamma, beta


def _forward_stats(av, w_dev, M, gamma, beta):
    C_, m = w_dev.shape
    d = lambda *sh, dt=torch.float32: torch.empty(*sh, dtype=dt, device=DEV)   # noqa: E731
    G, s = d(m, m, dt=torch.float64), d(m, dt=torch.float64)
    ops.gram(av, G, s, d(ops.gram_workspace_floats(M, m)))
    st = dict(scale=d(C_), shift=d(C_), mean=d(C_), invstd=d(C_), q=d(C_, m), G=G, s=s,
              rmean=torch.zeros(C_, device=DEV), rvar=torch.ones(C_, device=DEV))
    ops.bn_gram_finalize(G, s, w_dev, M, gamma.to(DEV), beta.to(DEV), st["rmean"], st["rvar"], st["scale"], st["shift"], st["mean"], st["invstd"], st["q"])
    return st


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 256, 0.0), (1, 20, 12, 128, 512, 3.0), (2, 8, 8, 256, 1024, 0.5), (2, 4, 4, 512, 2048, 0.0)])
def test_batchnorm_statistics_from_the_gram_matrix(case):
    """scale / shift / saved statistics / running statistics of BN(conv1x1(a)) without computing the conv - against the statistics of the
    float64 conv output (a large common mean makes E[y^2] - mean^2 cancel: f64 inside)"""
    B, H, W, m, C_, shift = case
    a, w, gamma, beta = _problem(B, H, W, m, C_, seed=m, mean_shift=shift)
    M = B * H * W
    st = _forward_stats(_dev_view(a), w.to(BF).to(DEV), M, gamma, beta)
    torch.cuda.synchronize()
    y = a.reshape(M, m).double() @ w.double().T
    mean, var = y.mean(0), y.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    close(st["mean"].cpu(), mean, 1e-5, "mean")
    close(st["invstd"].cpu(), invstd, 1e-4, "invstd")
    close(st["scale"].cpu(), gamma.double() * invstd, 1e-4, "scale")
    close(st["shift"].cpu(), beta.double() - mean * gamma.double() * invstd, 1e-4, "shift")
    close(st["rmean"].cpu(), 0.1 * mean, 1e-5, "running mean")
    close(st["rvar"].cpu(), 0.9 + 0.1 * var * M / (M - 1), 1e-4, "running var")
    close(st["q"].cpu(), w.double() @ (a.reshape(M, m).double().T @ a.reshape(M, m).double()), 1e-5, "W G")


@pytest.mark.parametrize("res_affine", [False, True])
@pytest.mark.parametrize("case", [(2, 16, 16, 64, 256), (1, 19, 13, 128, 128), (3, 8, 8, 256, 1024)])
def test_conv1x1_with_batchnorm_and_shortcut_in_the_epilogue(case, res_affine):
    """out = relu(scale * (a W^T) + shift + shortcut) in ONE launch (no stored conv output), the shortcut optionally through its own affine
    (the projection shortcut's BatchNorm) - against f64 from the same bf16 operands.  Two bf16 roundings (the BatchNorm output staged in
    LDS, then the stored sum - the unfused path rounded the stored conv output and the stored sum): 1.5 x 2^-8 of max, measured 1.1"""
    B, H, W, m, C_ = case
    a, w, gamma, beta = _problem(B, H, W, m, C_, seed=7 * m + C_)
    g = torch.Generator().manual_seed(1)
    r = _rq(torch.randn(B, H, W, C_, generator=g))
    sc, sh = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g)
    rs, rt = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g)
    out = ops.view(torch.zeros(B, H, W, C_ + 16, dtype=BF, device=DEV), 8, C_)
    wt = w.to(BF).to(DEV).view(1, C_, m)
    ops.conv2d(_dev_view(a), wt, None, out, 1, 1, relu=True, pre_affine=(sc.to(DEV), sh.to(DEV)), res=_dev_view(r, 16, 0),
               res_affine=(rs.to(DEV), rt.to(DEV)) if res_affine else None)
    torch.cuda.synchronize()
    y = a.double() @ w.double().T
    rr = r.double() * rs.double() + rt.double() if res_affine else r.double()
    ref = (y * sc.double() + sh.double() + rr).relu()
    close(out.torch().float().cpu(), ref, 1.5 * 2.0 ** -8, "fused tail")
    assert out.t[..., :8].abs().max().item() == 0 and out.t[..., 8 + C_:].abs().max().item() == 0, "wrote outside its channel range"


@pytest.mark.parametrize("case", [(2, 16, 16, 256, 64, 64), (1, 19, 13, 512, 128, 128), (2, 8, 8, 1024, 256, 256), (1, 24, 24, 128, 64, 128)])
def test_conv1x1_over_two_sources(case):
    """out[p] = W[:, :C1] x1[p] + W[:, C1:] x2[p] + bias - the second contraction source of cvcs_conv2d on the taps kernel (Cout % 128 == 0)
    and on the generic kernel (Cout = 64)"""
    B, H, W, C1, C2, N = case
    g = torch.Generator().manual_seed(C1 + N)
    x1, x2 = _rq(torch.randn(B, H, W, C1, generator=g)), _rq(torch.randn(B, H, W, C2, generator=g))
    w = _rq(torch.randn(N, C1 + C2, generator=g) / (C1 + C2) ** 0.5)
    bias = torch.randn(N, generator=g)
    out = ops.view(torch.zeros(B, H, W, N, dtype=BF, device=DEV))
    ops.conv2d(_dev_view(x1, 0, 32), w.to(BF).to(DEV).view(1, N, C1 + C2), bias.to(DEV), out, 1, 1, x2=_dev_view(x2, 64, 0))
    torch.cuda.synchronize()
    ref = torch.cat([x1, x2], -1).double() @ w.double().T + bias.double()
    close(out.torch().float().cpu(), ref, 2.0 ** -8, "two-source conv")


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 256), (1, 20, 12, 128, 512), (2, 8, 8, 256, 1024), (2, 4, 4, 512, 2048)])
def test_backward_of_conv1x1_batchnorm_without_the_conv_output(case):
    """z = BN(conv1x1(a)) (train mode), loss = sum(z * dz): dgamma, dbeta, dW and da of autograd in float64 from (a, dz) alone - the chain
    cvcs_gram -> cvcs_bn_gram_finalize -> weight-gradient GEMM (R = dz^T a, sum dz) -> cvcs_bn_gram_bwd -> cvcs_bn_gram_mmat -> ONE data-gradient
    GEMM over [dz | a]"""
    B, H, W, m, C_ = case
    a, w, gamma, beta = _problem(B, H, W, m, C_, seed=3 * m)
    g = torch.Generator().manual_seed(5)
    dz = _rq(torch.randn(B, H, W, C_, generator=g) * 1e-2)
    M = B * H * W
    # autograd, float64
    A = a.reshape(M, m).double().requires_grad_(True)
    Wt = w.double().requires_grad_(True)
    gam, bet = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = A @ Wt.T
    z = F.batch_norm(y, None, None, gam, bet, training=True, eps=1e-5)
    (z * dz.reshape(M, C_).double()).sum().backward()
    # device
    av, dzv = _dev_view(a, 0, 0), _dev_view(dz, 0, 32)
    w_dev = w.to(BF).to(DEV)
    st = _forward_stats(av, w_dev, M, gamma, beta)
    d = lambda *sh, dt=torch.float32: torch.empty(*sh, dtype=dt, device=DEV)   # noqa: E731
    R, sdz = d(C_, m), d(C_)
    need = ops.wgrad_workspace_floats_for(av, dzv, 1, 1, 1, 0, with_bias=True)
    assert ops.wgrad_takes_bias(av, dzv, 1, 1, 1, 0)
    ops.conv2d_wgrad(av, dzv, R, 1, 1, 1, 0, d(need), dbias=sdz)
    dgamma, dbeta, dw, coef = d(C_), d(C_), d(C_, m), d(2, C_)
    wd2, bias2 = torch.zeros(m, C_ + m, dtype=BF, device=DEV), d(m)
    ops.bn_gram_bwd(R, sdz, w_dev, st["q"], st["s"], M, st["scale"], st["mean"], st["invstd"], dgamma, dbeta, dw, wd2, coef)
    ops.bn_gram_mmat(w_dev, coef, wd2, bias2)
    da = ops.view(torch.zeros(B, H, W, m, dtype=BF, device=DEV))
    ops.conv2d(dzv, wd2.view(1, m, C_ + m), bias2, da, 1, 1, x2=av)
    torch.cuda.synchronize()
    close(dbeta.cpu(), bet.grad, 1e-5, "dbeta")
    close(dgamma.cpu(), gam.grad, 2e-4, "dgamma")
    close(dw.cpu(), Wt.grad, 2e-4, "dW")
    close(da.torch().float().cpu().reshape(M, m), A.grad, 2.0 ** -7, "da")


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 256), (1, 19, 13, 128, 512)])
def test_data_gradient_that_writes_the_previous_blocks_dz(case):
    """dz = (W x + g_shortcut) * (out > 0) in the epilogue of the 1x1 data-gradient launch (cvcs_conv_desc.res + mask): the tail backward of the
    block below without its pass over three tensors"""
    B, H, W, m, C_ = case
    g = torch.Generator().manual_seed(m)
    dy = _rq(torch.randn(B, H, W, m, generator=g))
    w = _rq(torch.randn(C_, m, generator=g) / m ** 0.5)
    gs = _rq(torch.randn(B, H, W, C_, generator=g))
    out_act = _rq(torch.randn(B, H, W, C_, generator=g).relu())      # half of it exact zeros
    dz = ops.view(torch.zeros(B, H, W, C_, dtype=BF, device=DEV))
    ops.conv2d(_dev_view(dy), w.to(BF).to(DEV).view(1, C_, m), None, dz, 1, 1, res=_dev_view(gs, 8, 0), mask=_dev_view(out_act, 0, 8))
    torch.cuda.synchronize()
    ref = (dy.double() @ w.double().T + gs.double()) * (out_act > 0)
    close(dz.torch().float().cpu(), ref, 1.5 * 2.0 ** -8, "fused tail backward")
    assert (dz.torch().float().cpu()[out_act == 0] == 0).all()


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 256), (1, 19, 13, 128, 512)])
def test_relu_mask_as_bits_written_by_the_forward_and_read_by_the_data_gradient(case):
    """cvcs_conv_desc.mask_bits_out / mask_bits: the forward launch of a block tail writes (out > 0) as one byte per 8 channels of a pixel; the
    consumer's tail-fused data gradient masked by those bits equals the launch masked by the output tensor itself, bit for bit"""
    B, H, W, m, C_ = case
    g = torch.Generator().manual_seed(m + 1)
    a = _rq(torch.randn(B, H, W, m, generator=g))
    w3 = _rq(torch.randn(C_, m, generator=g) / m ** 0.5)
    short = _rq(torch.randn(B, H, W, C_, generator=g))
    out = ops.view(torch.zeros(B, H, W, C_, dtype=BF, device=DEV))
    bits = torch.full((B * H * W * C_ // 8,), 0xa5, dtype=torch.uint8, device=DEV)
    one, zero = torch.ones(C_, device=DEV), torch.zeros(C_, device=DEV)
    ops.conv2d(_dev_view(a), w3.to(BF).to(DEV).view(1, C_, m), None, out, 1, 1, relu=True, pre_affine=(one, zero), res=_dev_view(short), mask_bits_out=bits)
    torch.cuda.synchronize()
    o = out.torch().float().cpu().reshape(-1, C_ // 8, 8)
    want = ((o > 0).to(torch.int32) << torch.arange(8, dtype=torch.int32)).sum(-1).to(torch.uint8)
    assert torch.equal(bits.cpu().view(-1, C_ // 8), want)
    assert 0.2 < (o > 0).float().mean() < 0.8
    # a plain two-source launch with a ReLU (the tail with a same-resolution projection shortcut) writes them too
    h = _rq(torch.randn(B, H, W, m, generator=g))
    w2 = torch.cat([w3, _rq(torch.randn(C_, m, generator=g) / m ** 0.5)], 1).contiguous()
    out2 = ops.view(torch.zeros(B, H, W, C_, dtype=BF, device=DEV))
    bits2 = torch.zeros_like(bits)
    ops.conv2d(_dev_view(a), w2.to(BF).to(DEV).view(1, C_, 2 * m), None, out2, 1, 1, relu=True, x2=_dev_view(h), mask_bits_out=bits2)
    torch.cuda.synchronize()
    o2 = out2.torch().float().cpu().reshape(-1, C_ // 8, 8)
    assert torch.equal(bits2.cpu().view(-1, C_ // 8), ((o2 > 0).to(torch.int32) << torch.arange(8, dtype=torch.int32)).sum(-1).to(torch.uint8))
    # the consumer: dz = (W dy + g_shortcut) * mask, from the tensor and from the bits
    dy = _rq(torch.randn(B, H, W, m, generator=g))
    gs = _rq(torch.randn(B, H, W, C_, generator=g))
    dz_t, dz_b = (ops.view(torch.zeros(B, H, W, C_, dtype=BF, device=DEV)) for _ in range(2))
    wd = w3.to(BF).to(DEV).view(1, C_, m)
    ops.conv2d(_dev_view(dy), wd, None, dz_t, 1, 1, res=_dev_view(gs), mask=out)
    ops.conv2d(_dev_view(dy), wd, None, dz_b, 1, 1, res=_dev_view(gs), mask_bits=bits)
    torch.cuda.synchronize()
    assert torch.equal(dz_t.torch(), dz_b.torch())


@pytest.mark.parametrize("half", [False, True])
def test_data_gradient_with_two_more_gradient_sources(half):
    """dz = (W x + g_a + g_b) * (out > 0) where g_b may live at half resolution (the data gradient of a 1x1 / stride-2 projection shortcut
    contributes at the even pixels only): cvcs_conv_desc.res + res2 + mask - the tail backward below a down-sampling block in one launch"""
    B, H, W, m, C_ = 2, 16, 24, 128, 256
    g = torch.Generator().manual_seed(11 + half)
    dy = _rq(torch.randn(B, H, W, m, generator=g))
    w = _rq(torch.randn(C_, m, generator=g) / m ** 0.5)
    ga = _rq(torch.randn(B, H, W, C_, generator=g))
    gb = _rq(torch.randn(B, H // 2, W // 2, C_, generator=g)) if half else _rq(torch.randn(B, H, W, C_, generator=g))
    out_act = _rq(torch.randn(B, H, W, C_, generator=g).relu())
    dz = ops.view(torch.zeros(B, H, W, C_, dtype=BF, device=DEV))
    ops.conv2d(_dev_view(dy), w.to(BF).to(DEV).view(1, C_, m), None, dz, 1, 1, res=_dev_view(ga, 8, 0), res2=(_dev_view(gb, 0, 16), half),
               mask=_dev_view(out_act))
    torch.cuda.synchronize()
    tot = dy.double() @ w.double().T + ga.double()
    if half:
        tot[:, ::2, ::2, :] += gb.double()
    else:
        tot = tot + gb.double()
    close(dz.torch().float().cpu(), tot * (out_act > 0), 1.5 * 2.0 ** -8, "fused tail backward, three sources")
