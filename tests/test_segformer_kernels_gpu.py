"""-m gpu: the kernels of csrc/segformer.hip through the C-ABI against plain torch (f32 and bf16 storage)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from cvcs_amd import ops
    return ops


def _nhwc(x):       # [B,C,H,W] -> contiguous [B,H,W,C]
    return x.permute(0, 2, 3, 1).contiguous()


def _tol(dt):
    return dict(atol=1e-5, rtol=1e-5) if dt == torch.float32 else dict(atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C,k,s,p", [(2, 16, 16, 32, 3, 2, 1), (2, 16, 24, 64, 8, 8, 0), (1, 12, 12, 160, 2, 2, 0), (2, 9, 7, 8, 3, 1, 1)])
def test_im2col_is_unfold_and_col2im_its_transpose(dt, B, H, W, C, k, s, p):
    ops = _ops()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, C, H, W, generator=g).to(dt)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    Kp = -(-k * k * C // 32) * 32
    xv = ops.view(_nhwc(x).to(DEV))
    col = torch.full((B, Ho, Wo, Kp), 7.0, dtype=dt, device=DEV)
    ops.im2col(xv, ops.view(col), k, k, s, -p, -p)
    # torch.unfold orders the patch (c, ky, kx); ours is (ky, kx, c)
    ref = F.unfold(x.float(), k, padding=p, stride=s).view(B, C, k * k, Ho * Wo).permute(0, 3, 2, 1).reshape(B, Ho, Wo, k * k * C)
    assert torch.equal(col[..., :k * k * C].float().cpu(), ref)
    assert (col[..., k * k * C:] == 0).all()
    # the transpose: fold of a random patch matrix
    gcol = torch.randn(B, Ho, Wo, Kp, generator=g).to(dt)
    dx = torch.empty(B, H, W, C, dtype=dt, device=DEV)
    ops.col2im(ops.view(gcol.to(DEV)), ops.view(dx), k, k, s, -p, -p)
    gref = gcol[..., :k * k * C].float().view(B, Ho * Wo, k * k, C).permute(0, 3, 2, 1).reshape(B, C * k * k, Ho * Wo)
    want = _nhwc(F.fold(gref, (H, W), k, padding=p, stride=s))
    assert torch.allclose(dx.float().cpu(), want, **_tol(dt))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("u8", [True, False])
def test_stem_patches_carry_the_reference_normalisation(dt, u8):
    ops = _ops()
    B, H, W, k, s, p = 2, 32, 40, 7, 4, 3
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 256, (B, 3, H, W), generator=g, dtype=torch.uint8)
    xin = x if u8 else x.float()
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    col = torch.empty(B, Ho, Wo, 160, dtype=dt, device=DEV)
    ops.im2col_stem(xin.to(DEV), ops.view(col), k, k, s, p)
    mean = torch.tensor(ops.SEGFORMER_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(ops.SEGFORMER_STD).view(1, 3, 1, 1)
    xn = ((x.float() - mean) / std).to(dt).float()
    ref = F.unfold(xn, k, padding=p, stride=s).view(B, 3, k * k, Ho * Wo).permute(0, 3, 2, 1).reshape(B, Ho, Wo, 147)
    assert torch.equal(col[..., :147].float().cpu(), ref)
    assert (col[..., 147:] == 0).all()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k,p", [(8, 3), (4, 1)])
def test_conv_transpose_as_four_phase_patch_matrices(dt, k, p):
    """ConvTranspose2d(C, C, k, stride 2, padding p) (S/nets.py:331-334) = per output parity (a, b) a GEMM of the phase's patch matrix with
    the taps w[ci][co][((a+p)&1) + 2 jy][((b+p)&1) + 2 jx]; the data gradient = col2im over the four phases"""
    ops = _ops()
    B, H, W, C = 2, 6, 5, 16
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, C, H, W, generator=g).to(dt)
    w = (torch.randn(C, C, k, k, generator=g) * 0.2).to(dt)
    k2 = k // 2
    K = k2 * k2 * C
    xv = ops.view(_nhwc(x).to(DEV))
    cols = torch.empty(4, B, H, W, K, dtype=dt, device=DEV)
    outs = torch.empty(4, B, H, W, C, dtype=dt, device=DEV)
    oy0 = [(a + p) >> 1 for a in (0, 0, 1, 1)]
    ox0 = [(b + p) >> 1 for b in (0, 1, 0, 1)]
    for ph in range(4):
        a, b = ph >> 1, ph & 1
        ops.im2col(xv, ops.view(cols[ph]), k2, k2, 1, oy0[ph], ox0[ph], -1)
        wp = w.float()[:, :, ((a + p) & 1)::2, ((b + p) & 1)::2]                      # [ci, co, jy, jx]
        wmat = wp.permute(2, 3, 0, 1).reshape(K, C).to(DEV)                           # [(jy, jx, ci), co]
        outs[ph] = (cols[ph].float().view(-1, K) @ wmat).view(B, H, W, C).to(dt)
    full = torch.empty(B, 2 * H, 2 * W, C, dtype=dt, device=DEV)
    ops.phase_shuffle(ops.view(outs[0]), outs[0].numel(), ops.view(full), relu=True)
    want = _nhwc(F.relu(F.conv_transpose2d(x.float(), w.float(), stride=2, padding=p)))
    assert torch.allclose(full.float().cpu(), want, **(_tol(dt) if dt == torch.float32 else dict(atol=6e-2, rtol=3e-2)))
    # backward: unshuffle with the ReLU mask, then the transposed gather over the four phases
    gout = torch.randn(B, 2 * H, 2 * W, C, generator=g).to(dt).to(DEV)
    gph = torch.empty_like(outs)
    ops.phase_unshuffle(ops.view(gout), ops.view(full), ops.view(gph[0]), gph[0].numel())
    masked = gout.float() * (full.float() > 0)
    for ph in range(4):
        a, b = ph >> 1, ph & 1
        assert torch.equal(gph[ph].float(), masked[:, a::2, b::2, :])
    gcols = torch.randn(4, B, H, W, K, generator=g).to(dt).to(DEV)
    dx = torch.empty(B, H, W, C, dtype=dt, device=DEV)
    ops.col2im(ops.view(gcols[0]), ops.view(dx), k2, k2, 1, oy0, ox0, -1, phases=4, phase_elems=gcols[0].numel())
    # reference: the adjoint of the four gathers (col[oy, ox, tap] = x[oy + oy0 - jy, ox + ox0 - jx]), tap by tap
    want = torch.zeros(B, H, W, C)
    gc = gcols.float().cpu()
    for ph in range(4):
        for jy in range(k2):
            for jx in range(k2):
                dy_, dx_ = jy - oy0[ph], jx - ox0[ph]
                iy0, iy1, ix0, ix1 = max(0, -dy_), min(H, H - dy_), max(0, -dx_), min(W, W - dx_)
                if iy0 < iy1 and ix0 < ix1:
                    t = jy * k2 + jx
                    want[:, iy0:iy1, ix0:ix1] += gc[ph][:, iy0 + dy_:iy1 + dy_, ix0 + dx_:ix1 + dx_, t * C:(t + 1) * C]
    assert torch.allclose(dx.float().cpu(), want, **(_tol(dt) if dt == torch.float32 else dict(atol=8e-2, rtol=3e-2)))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 128), (1, 7, 9, 640), (2, 4, 4, 1024), (2, 8, 8, 160)])
def test_depthwise_3x3_forward_data_and_weight_gradients(dt, B, H, W, C):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, C, H, W, generator=g).to(dt)
    w = torch.randn(C, 1, 3, 3, generator=g) * 0.3
    bias = torch.randn(C, generator=g) * 0.1
    xv = ops.view(_nhwc(x).to(DEV))
    out = torch.empty(B, H, W, C, dtype=dt, device=DEV)
    ops.dwconv3x3(xv, w.to(DEV), bias.to(DEV), ops.view(out))
    xr = x.float().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, padding=1, groups=C)
    assert torch.allclose(out.float().cpu(), _nhwc(ref.detach()), **_tol(dt))
    gy = torch.randn(B, C, H, W, generator=g).to(dt)
    ref.backward(gy.float())
    gyv = ops.view(_nhwc(gy).to(DEV))
    dx = torch.empty_like(out)
    ops.dwconv3x3(gyv, w.to(DEV), None, ops.view(dx), flip=True)
    assert torch.allclose(dx.float().cpu(), _nhwc(xr.grad), **_tol(dt))
    rows = ops.dwconv3x3_wgrad_rows(B * H * W)
    part = torch.empty(rows, 10 * C, dtype=torch.float32, device=DEV)
    ops.dwconv3x3_wgrad(xv, gyv, part)
    gwb = torch.empty(10 * C, dtype=torch.float32, device=DEV)
    ops.colsum_finalize(part, rows, 10 * C, gwb)
    scale = max(1.0, wr.grad.abs().max().item())
    assert torch.allclose(gwb[:9 * C].cpu().view(C, 1, 3, 3), wr.grad, atol=(1e-4 if dt == torch.float32 else 1e-3) * scale, rtol=1e-4)
    assert torch.allclose(gwb[9 * C:].cpu(), br.grad, atol=1e-4 * max(1.0, br.grad.abs().max().item()), rtol=1e-4)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,Hq,Hk,heads,D", [(2, 16, 2, 1, 32), (2, 8, 2, 2, 32), (1, 20, 10, 5, 32), (2, 4, 4, 8, 32), (1, 32, 18, 1, 32), (2, 32, 8, 2, 32),
                                              (1, 40, 16, 5, 32), (1, 24, 24, 1, 32), (2, 16, 2, 1, 64), (1, 24, 8, 2, 64), (1, 20, 16, 5, 64), (1, 8, 18, 8, 64)])
def test_spatial_reduction_attention_forward_and_backward(dt, B, Hq, Hk, heads, D):
    """softmax(q k^T / sqrt(D)) v (D = 32: MiT-b0, 64: MiT-b1 ... b5) with N = Hq^2 queries and Nk = Hk^2 keys per image and head (Nk = 4 ... 576: below one 8-key block,
    across the 256-key LDS chunk; Nk = 64 / 256 / 576 in bf16 run the matrix-core kernels, ragged query counts included), gradients
    against autograd"""
    ops = _ops()
    C = heads * D
    N, Nk = Hq * Hq, Hk * Hk
    g = torch.Generator().manual_seed(4)
    q = torch.randn(B, N, C, generator=g).to(dt)
    kv = torch.randn(B, Nk, 2 * C, generator=g).to(dt)
    go = torch.randn(B, N, C, generator=g).to(dt)
    qr, kvr = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    qh = qr.view(B, N, heads, D).transpose(1, 2)
    kh = kvr[..., :C].reshape(B, Nk, heads, D).transpose(1, 2)
    vh = kvr[..., C:].reshape(B, Nk, heads, D).transpose(1, 2)
    att = torch.softmax(qh @ kh.transpose(2, 3) / math.sqrt(D), dim=-1)
    ref = (att @ vh).transpose(1, 2).reshape(B, N, C)
    ref.backward(go.float())
    qv = ops.view(q.view(B, Hq, Hq, C).to(DEV))
    kvv = ops.view(kv.view(B, Hk, Hk, 2 * C).to(DEV))
    out = torch.empty(B, Hq, Hq, C, dtype=dt, device=DEV)
    lse = torch.empty(B * heads * N, dtype=torch.float32, device=DEV)
    ops.sr_attention_fwd(qv, kvv, heads, ops.view(out), lse)
    tol = dict(atol=2e-5, rtol=1e-4) if dt == torch.float32 else dict(atol=2e-2, rtol=2e-2)
    assert torch.allclose(out.float().cpu().view(B, N, C), ref.detach(), **tol)
    want_lse = torch.logsumexp(qh @ kh.transpose(2, 3) / math.sqrt(D), dim=-1).detach().reshape(-1)
    assert torch.allclose(lse.cpu(), want_lse, atol=1e-4, rtol=1e-4)
    gq = torch.empty_like(out)
    gkv = torch.empty(B, Hk, Hk, 2 * C, dtype=dt, device=DEV)
    ws = torch.empty(ops.sr_attention_bwd_workspace(B, N, Nk, heads, D), dtype=torch.float32, device=DEV)
    # the backward sees the forward's stored output (bf16-rounded on that path)
    ops.sr_attention_bwd(qv, kvv, ops.view(out), ops.view(go.view(B, Hq, Hq, C).to(DEV)), lse, heads, ops.view(gq), ops.view(gkv), ws)
    gtol = dict(atol=5e-5, rtol=1e-3) if dt == torch.float32 else dict(atol=6e-2, rtol=3e-2)
    assert torch.allclose(gq.float().cpu().view(B, N, C), qr.grad, **gtol)
    sc = max(1.0, kvr.grad.abs().max().item())
    assert (gkv.float().cpu().view(B, Nk, 2 * C) - kvr.grad).abs().max().item() <= (1e-4 if dt == torch.float32 else 2e-2) * sc


def test_drop_path_scales_replay_and_scaled_residual_add():
    ops = _ops()
    B, n = 8, 6
    rates = [0.0, 0.05, 0.1, 0.5, 0.9, 0.1]
    state = torch.tensor([4242, 7], dtype=torch.int64, device=DEV)
    out = torch.empty(n, B, dtype=torch.float32, device=DEV)
    ops.drop_path_scales(state, torch.tensor(rates, device=DEV), B, out)
    host = ops.drop_path_scales_host(4242, 7, rates, B)
    assert torch.equal(out.cpu(), host)
    assert (host[0] == 1).all() and set(host[3].tolist()) <= {0.0, 2.0} and 0 < (host[3] == 0).sum() < B
    ops.counter_add(state, 1)
    ops.drop_path_scales(state, torch.tensor(rates, device=DEV), B, out)
    assert not torch.equal(out.cpu(), host)                      # the next step draws again
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(B, 5, 3, 32).to(dt).to(DEV)
        r = torch.randn(B, 5, 3, 32).to(dt).to(DEV)
        y = torch.empty_like(x)
        s = host[3].to(DEV)
        ops.scale_rows_add(ops.view(x), s, ops.view(r), ops.view(y))
        want = (r.float() + s.view(B, 1, 1, 1) * x.float()).to(dt)
        assert torch.equal(y, want)
        ops.scale_rows_add(ops.view(x), s, None, ops.view(y))
        assert torch.equal(y, (s.view(B, 1, 1, 1) * x.float()).to(dt))
        ops.scale_rows_add(ops.view(x), None, ops.view(r), ops.view(y))
        assert torch.equal(y, (r.float() + x.float()).to(dt))


def test_weight_gather_table_and_its_transpose():
    """Conv2d OIHW -> [Cout_pad][(ky, kx, ci) padded] and [(ky, kx, ci)][Cout_pad]; ConvTranspose2d IOHW phase (1, 0) of k = 8, p = 3;
    the scatter puts a gathered-layout gradient back into the master layout"""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    Co, Ci, k = 24, 8, 3
    w = torch.randn(Co, Ci, k, k, generator=g).to(DEV)
    K, Kp, Cop = k * k * Ci, 96, 32
    wf = torch.full((Cop, Kp), 9.0, dtype=torch.bfloat16, device=DEV)
    wd = torch.full((Kp, Cop), 9.0, dtype=torch.bfloat16, device=DEV)
    T = k * k
    items = [dict(src=w, dst=wf, R=Cop, Cp=Kp, Rv=Co, Cv=K, row=(1, 1 << 20, 0, 0, Ci * T), col=(1, Ci, 1, 0, T)),
             dict(src=w, dst=wd, R=Kp, Cp=Cop, Rv=K, Cv=Co, row=(1, Ci, 1, 0, T), col=(1, 1 << 20, 0, 0, Ci * T))]
    # ConvTranspose2d [Ci, Co, 8, 8], phase (a, b) = (1, 0), p = 3: rows co, columns (jy, jx, ci) -> w[ci][co][ay + 2 jy][ax + 2 jx]
    C, kk, p, a, b = 16, 8, 3, 1, 0
    wt = torch.randn(C, C, kk, kk, generator=g).to(DEV)
    ay, ax = (a + p) & 1, (b + p) & 1
    k2 = kk // 2
    wph = torch.empty(C, k2 * k2 * C, dtype=torch.float32, device=DEV)
    items.append(dict(src=wt, dst=wph, base=ay * kk + ax, R=C, Cp=k2 * k2 * C, Rv=C, Cv=k2 * k2 * C, row=(1, 1 << 20, 0, 0, kk * kk),
                      col=(k2, C, 2 * kk, 2, C * kk * kk), f32_out=True))
    table, n = ops.gather_table(items, DEV)
    ops.gather_weights(table, n, torch.bfloat16)
    ref_f = torch.zeros(Cop, Kp, device=DEV)
    ref_f[:Co, :K] = w.permute(0, 2, 3, 1).reshape(Co, K)
    assert torch.equal(wf.float(), ref_f.bfloat16().float())
    ref_d = torch.zeros(Kp, Cop, device=DEV)
    ref_d[:K, :Co] = w.permute(2, 3, 1, 0).reshape(K, Co)
    assert torch.equal(wd.float(), ref_d.bfloat16().float())
    want = wt[:, :, ay::2, ax::2].permute(1, 2, 3, 0).reshape(C, k2 * k2 * C)
    assert torch.equal(wph, want)
    # transpose: a gradient in the gathered layout back into a master-shaped buffer
    gmaster = torch.zeros_like(wt)
    ggath = torch.randn(C, k2 * k2 * C, generator=g).to(DEV)
    t2, n2 = ops.gather_table([dict(src=gmaster, dst=ggath, base=ay * kk + ax, R=C, Cp=k2 * k2 * C, Rv=C, Cv=k2 * k2 * C, row=(1, 1 << 20, 0, 0, kk * kk),
                                    col=(k2, C, 2 * kk, 2, C * kk * kk), f32_out=True)], DEV)
    ops.scatter_weight_grads(t2, n2)
    back = gmaster[:, :, ay::2, ax::2].permute(1, 2, 3, 0).reshape(C, k2 * k2 * C)
    assert torch.equal(back, ggath)
    assert (gmaster[:, :, 1 - ay::2, :] == 0).all()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k,p,layout", [(4, 1, 0), (8, 3, 1)])
def test_conv_transpose_as_one_convolution_and_its_gradient_unpack(dt, k, p, layout):
    """cvcs_deconv_pack: ConvTranspose2d(NC, NC, k, 2, p) == conv (k/2+1)^2 / stride 1 to 4 CP phase-major channels + pixel shuffle, for the
    conv-kernel layout (k = 4) and the patch-GEMM layout (k = 8); cvcs_deconv_unpack_grad returns autograd's weight / bias gradients"""
    ops = _ops()
    NC, CP, B, H, W = 5, 32, 2, 8, 8
    T = k // 2 + 1
    g = torch.Generator().manual_seed(6)
    w = (torch.randn(NC, NC, k, k, generator=g) * 0.3)
    bias = torch.randn(NC, generator=g) * 0.1
    x = torch.randn(B, NC, H, W, generator=g).to(dt)
    xp = torch.zeros(B, H, W, CP, dtype=dt)
    xp[..., :NC] = _nhwc(x)
    kg = 32 if dt == torch.bfloat16 else 16
    if layout == 0:
        wf = torch.zeros(T * T, 4 * CP, max(CP, kg), dtype=dt, device=DEV)
        wd = torch.zeros(T * T, CP, 4 * CP, dtype=dt, device=DEV)
    else:
        Kp = -(-T * T * CP // 128) * 128
        wf = torch.zeros(1, 4 * CP, Kp, dtype=dt, device=DEV)
        wd = torch.zeros(1, Kp, 4 * CP, dtype=dt, device=DEV)
    bias4 = torch.zeros(4 * CP, device=DEV)
    ops.deconv_pack(w.to(DEV), bias.to(DEV), k, p, CP, layout, wf, wd, bias4)
    xv = ops.view(xp.to(DEV))
    y4 = torch.empty(B, H, W, 4 * CP, dtype=dt, device=DEV)
    if layout == 0:
        ops.conv2d(xv, wf, bias4, ops.view(y4), T, T, 1, T // 2)
    else:
        col = torch.empty(B, H, W, wf.shape[2], dtype=dt, device=DEV)
        ops.im2col(xv, ops.view(col), T, T, 1, -(T // 2), -(T // 2))
        ops.conv2d(ops.view(col), wf, bias4, ops.view(y4), 1, 1)
    full = torch.empty(B, 2 * H, 2 * W, CP, dtype=dt, device=DEV)
    ops.phase_shuffle(ops.view(y4), CP, ops.view(full), relu=False)
    wr = w.to(dt).float().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    xr = x.float().requires_grad_(True)
    ref = F.conv_transpose2d(xr, wr, br, stride=2, padding=p)
    tol = dict(atol=2e-5, rtol=1e-4) if dt == torch.float32 else dict(atol=5e-2, rtol=3e-2)
    assert torch.allclose(full[..., :NC].float().cpu(), _nhwc(ref.detach()), **tol)
    assert (full[..., NC:] == 0).all()
    # backward through the same operands: dgrad with wd, weight gradient + unpack
    go = torch.randn(B, NC, 2 * H, 2 * W, generator=g).to(dt)
    ref.backward(go.float())
    gp = torch.zeros(B, 2 * H, 2 * W, CP, dtype=dt)
    gp[..., :NC] = _nhwc(go)
    g4 = torch.empty(B, H, W, 4 * CP, dtype=dt, device=DEV)
    ops.phase_unshuffle(ops.view(gp.to(DEV)), None, ops.view(g4), CP)
    gx = torch.empty(B, H, W, CP, dtype=dt, device=DEV)
    if layout == 0:
        gw = torch.zeros(4 * CP, CP, T, T, device=DEV)
        ws = torch.empty(ops.wgrad_workspace_floats_for(xv, ops.view(g4), T, T, 1, T // 2), device=DEV)
        ops.conv2d_wgrad(xv, ops.view(g4), gw, T, T, 1, T // 2, ws)
        ops.conv2d(ops.view(g4), wd, None, ops.view(gx), T, T, 1, T // 2)
    else:
        gw = torch.zeros(4 * CP, wf.shape[2], 1, 1, device=DEV)
        ws = torch.empty(ops.wgrad_workspace_floats_for(ops.view(col), ops.view(g4), 1, 1, 1, 0), device=DEV)
        ops.conv2d_wgrad(ops.view(col), ops.view(g4), gw, 1, 1, 1, 0, ws)
        gcol = torch.empty_like(col)
        ops.conv2d(ops.view(g4), wd, None, ops.view(gcol), 1, 1)
        ops.col2im(ops.view(gcol), ops.view(gx), T, T, 1, -(T // 2), -(T // 2))
    gtol = dict(atol=1e-4, rtol=1e-3) if dt == torch.float32 else dict(atol=1e-1, rtol=5e-2)
    assert torch.allclose(gx[..., :NC].float().cpu(), _nhwc(xr.grad), **gtol)
    db4 = g4.float().sum(dim=(0, 1, 2))
    gwm, gbm = torch.zeros(NC, NC, k, k, device=DEV), torch.zeros(NC, device=DEV)
    ops.deconv_unpack_grad(gw, db4, k, p, CP, layout, gwm, gbm)
    sc = wr.grad.abs().max().item()
    assert (gwm.cpu() - wr.grad).abs().max().item() <= (1e-4 if dt == torch.float32 else 3e-2) * sc
    assert torch.allclose(gbm.cpu(), br.grad, atol=(1e-3 if dt == torch.float32 else 2e-1), rtol=1e-3)

