"""Python-side call wrappers over the C-ABI: torch only supplies device memory and the current HIP stream.

An activation is a `View`: a channel range [off, off+C) of an NHWC torch tensor [B, H, W, ld].  Nothing here
computes on the host; every function is one (or a few) asynchronous launches on torch's current stream.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib
from ._lib import BF16, F32, ConvDesc, WgradDesc, check

TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16}
KGROUP = {F32: 16, BF16: 32}  # channels per 64-byte K-step of the conv kernel


def dtype_code(t: torch.dtype) -> int:
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported activation dtype {t}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class KernelTimers:
    """HIP-event brackets around individual launches ON THE STREAM THEY ARE LAUNCHED ON (torch's current stream is the
    stream handed to the C-ABI).  bench.py switches this on for the timed region to measure the dominant kernels'
    average launch duration; off (None) by default, so the product path records nothing."""

    class _Off:
        def record(self, *a):
            pass
    _OFF = (_Off(), _Off())

    def __init__(self, only=None):
        self.records = {}   # kind -> list of (start_event, end_event, algorithmic_flops)
        self.only = only    # None: every tagged launch; else a set of kernel families ("family[:scope]" kinds; the others are not bracketed)

    def bracket(self, kind, flops):
        if self.only is not None and kind.split(":")[0] not in self.only:
            return self._OFF
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.setdefault(kind, []).append((s, e, flops))
        return s, e

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in recs)
            fl = sum(f for _, _, f in recs)
            out[kind] = dict(launches=len(recs), total_ms=ms, avg_us=1e3 * ms / len(recs), flops=fl,
                             tflops=fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)
        return out


TIMERS: KernelTimers | None = None
SCOPE = ""   # set by the engines around a layer's launches: timer kinds become "<kernel family>:<scope>" (enc | dec)


def _ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


@dataclass
class View:
    t: torch.Tensor  # [B, H, W, ld], contiguous
    off: int
    C: int

    @property
    def B(self): return self.t.shape[0]
    @property
    def H(self): return self.t.shape[1]
    @property
    def W(self): return self.t.shape[2]
    @property
    def ld(self): return self.t.shape[3]
    @property
    def ptr(self): return self.t.data_ptr() + self.off * self.t.element_size()
    @property
    def code(self): return dtype_code(self.t.dtype)

    def torch(self):
        return self.t[..., self.off:self.off + self.C]


class UpView(View):
    """the bilinear x2 up-sampling of a view that is never stored: H and W are those of the up-sampled map, ptr / ld those of the map at half the
    resolution (cvcs_conv_desc.in_up2, cvcs_wgrad_desc.x_up2: the thin 3x3 kernels interpolate while they stage their input)"""
    @property
    def H(self): return 2 * self.t.shape[1]
    @property
    def W(self): return 2 * self.t.shape[2]


def up_view(v: View) -> UpView:
    return UpView(v.t, v.off, v.C)


def view(t: torch.Tensor, off: int = 0, C_: int | None = None) -> View:
    assert t.dim() == 4 and t.is_contiguous()
    return View(t, off, t.shape[3] - off if C_ is None else C_)


def new_act(B, H, W, C_, dtype, device):
    return torch.empty((B, H, W, C_), dtype=dtype, device=device)


# ------------------------------------------------------------------------------------------------ convolution
def conv_out_hw(H, W, KH, KW, stride, pad, dil=1):
    return ((H + 2 * pad - dil * (KH - 1) - 1) // stride + 1, (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1)


def _conv_desc(x: View, wt, bias, out: View, KH, KW, stride, pad, dil, relu, pixel_shuffle, stats, virt=None):
    Cout = wt.shape[1]
    d = ConvDesc()
    if virt is not None:
        # virtual-pixel input (the 7x7/s2 stem as a 7x1 filter over (kw, c) pairs, cvcs_conv_desc.aniso): `x` is the
        # [B, H, W+8, 4] tensor of ops.pack_input_stem; a virtual pixel = 32 elements starting every 8
        assert (KH, KW, stride, pad, dil) == (7, 1, 2, 3, 1) and x.C == 4 and x.off == 0 and wt.shape[2] == 32
        Wv = (x.W - 8) // 2
        Ho, Wo = conv_out_hw(x.H, Wv, 7, 1, 2, 3)[0], Wv
        d.in_, d.in_ld, d.B, d.H, d.W, d.Cin = x.ptr, 8, x.B, x.H, Wv, 32
        d.aniso, d.stride_w, d.pad_w = 1, 1, 0
        d.in_row_pitch, d.in_img_pitch = x.W * 4, x.H * x.W * 4
    else:
        Ho, Wo = conv_out_hw(x.H, x.W, KH, KW, stride, pad, dil)
        d.in_, d.in_ld, d.B, d.H, d.W, d.Cin = x.ptr, x.ld, x.B, x.H, x.W, x.C
    if isinstance(x, UpView):
        d.in_up2 = 1
    if virt is None and wt.shape[2] > x.C:     # narrow input (the 3-channel tile stored 16 bytes per pixel) under a zero-padded K-group
        d.Cin, d.Cin_valid = wt.shape[2], x.C
    d.wt, d.bias = _ptr(wt), _ptr(bias)
    d.out, d.out_ld, d.Ho, d.Wo, d.Cout = 0 if out is None else out.ptr, 0 if out is None else out.ld, Ho, Wo, Cout
    d.KH, d.KW, d.stride, d.pad, d.dil = KH, KW, stride, pad, dil
    d.relu, d.pixel_shuffle = int(relu), int(pixel_shuffle)
    if stats is not None:
        d.stat_sum, d.stat_m2, d.stat_cnt = (_ptr(t) for t in stats)
    d.dtype = x.code
    return d, Ho, Wo, Cout


class _Shape:
    """shape-only stand-in for a weight tensor ([taps][Cout][Cin]) when only the statistics row count is wanted"""
    def __init__(self, taps, cout, cin):
        self.shape = (taps, cout, cin)

    def data_ptr(self):
        return 0


def conv_stat_rows(x: View, Cout, KH, KW, stride=1, pad=0, dil=1, virt=None, bwd=False) -> int:
    """number of partial-statistics rows cvcs_conv2d writes for this geometry (bwd: of partial rows of the fused BatchNorm-backward reduce,
    cvcs_conv_desc.bwd_y - the thin kernel writes one per workgroup)"""
    # (the weight image's contraction axis is at least one K-group wide: a 16-channel bf16 input runs under a zero-padded 32-channel K-group)
    d, _, _, _ = _conv_desc(x, _Shape(KH * KW, Cout, 32 if virt else max(x.C, KGROUP[x.code])), None, None, KH, KW, stride, pad, dil, False, False, None,
                            virt=virt)
    if bwd:
        d.bwd_y, d.bwd_mode = 16, 0      # (a non-null marker: the query reads no tensor)
    n = _lib.lib().cvcs_conv_stat_rows(C.byref(d))
    if n < 0:
        raise _lib.CvcsError("cvcs_conv_stat_rows: bad shape")
    return n


def conv2d(x: View, wt: torch.Tensor, bias, out: View, KH, KW, stride=1, pad=0, dil=1, relu=False,
           pixel_shuffle=False, stats=None, cin_real=None, pre_affine=None, post_affine=None, pool: View | None = None,
           bn_bwd=None, virt=None, flops=None, res: View | None = None, res_affine=None, x2: View | None = None, mask: View | None = None, res2=None,
           mask_bits: torch.Tensor | None = None, mask_bits_out: torch.Tensor | None = None):
    """out = act(conv(x, wt) + bias); wt packed [KH*KW][Cout][Cin]; out.C = Cout (or Cout/4 when pixel_shuffle).
    stats = (sum[rows*Cout], m2[rows*Cout], cnt[rows]) f32 buffers, rows = conv_stat_rows(...).
    bn_bwd = (y view, scale, shift, mean, invstd, mode, part_dz, part_dzx): the launch also takes the first pass of the
    BatchNorm backward that consumes `out` (bf16 3x3 / stride 1 / pad 1; one partial row per 16x16 tile)."""
    d, Ho, Wo, Cout = _conv_desc(x, wt, bias, out, KH, KW, stride, pad, dil, relu, pixel_shuffle, stats, virt=virt)
    if res is not None:
        # residual epilogue (bf16 1x1): out = relu?(pre_scale * acc + pre_shift + (res | res_scale * res + res_shift))
        assert (res.B, res.H, res.W, res.C) == (x.B, Ho, Wo, Cout) and res.t.dtype == x.t.dtype
        d.res, d.res_ld = res.ptr, res.ld
        if res_affine is not None:
            d.res_scale, d.res_shift = res_affine[0].data_ptr(), res_affine[1].data_ptr()
    if res2 is not None:
        # a second residual tile, (view, half_resolution): the half-resolution one contributes at the even pixels only
        r2, half = res2
        want = (x.B, Ho // 2, Wo // 2, Cout) if half else (x.B, Ho, Wo, Cout)
        assert (r2.B, r2.H, r2.W, r2.C) == want and r2.t.dtype == x.t.dtype
        d.res2, d.res2_ld, d.res2_half = r2.ptr, r2.ld, int(half)
    if mask is not None:
        # ReLU backward in the epilogue: out = value * (mask > 0)
        assert (mask.B, mask.H, mask.W, mask.C) == (x.B, Ho, Wo, Cout) and mask.t.dtype == x.t.dtype
        d.mask, d.mask_ld = mask.ptr, mask.ld
    if mask_bits is not None:
        # the same mask as bits (one byte per 8 channels of a pixel, written by the forward launch: mask_bits_out)
        assert mask is None and mask_bits.dtype == torch.uint8 and mask_bits.is_contiguous() and mask_bits.numel() == x.B * Ho * Wo * Cout // 8
        d.mask_bits = mask_bits.data_ptr()
    if mask_bits_out is not None:
        assert relu and mask_bits_out.dtype == torch.uint8 and mask_bits_out.is_contiguous() and mask_bits_out.numel() == x.B * Ho * Wo * Cout // 8
        d.mask_bits_out = mask_bits_out.data_ptr()
    if x2 is not None:
        # second contraction source of the same pixels (bf16 1x1): wt rows are x.C + x2.C long
        assert (x2.B, x2.H, x2.W) == (x.B, x.H, x.W) and x2.t.dtype == x.t.dtype and wt.shape[2] == x.C + x2.C
        d.in2, d.in2_ld, d.Cin2 = x2.ptr, x2.ld, x2.C
        d.Cin, d.Cin_valid = x.C, 0          # (_conv_desc read the longer weight rows as a zero-padded K-group)
    if bn_bwd is not None:
        yv, bsc, bsh, bmu, bis, bmode, p0, p1 = bn_bwd
        rows = conv_stat_rows(x, Cout, KH, KW, stride, pad, bwd=bmode == 0)
        assert (yv.B, yv.H, yv.W, yv.C) == (x.B, Ho, Wo, Cout) and yv.t.dtype == x.t.dtype
        assert p0.numel() >= rows * Cout and p1.numel() >= rows * Cout and p0.dtype == torch.float32
        d.bwd_y, d.bwd_y_ld = yv.ptr, yv.ld
        d.bwd_scale, d.bwd_shift, d.bwd_mean, d.bwd_invstd = bsc.data_ptr(), bsh.data_ptr(), bmu.data_ptr(), bis.data_ptr()
        d.bwd_mode = bmode
        d.bwd_part_dz, d.bwd_part_dzx = p0.data_ptr(), p1.data_ptr()
    # eval-mode BatchNorm folded into the epilogue: (scale, shift) before / after the ReLU
    if pre_affine is not None:
        d.pre_scale, d.pre_shift = pre_affine[0].data_ptr(), pre_affine[1].data_ptr()
    if post_affine is not None:
        d.post_scale, d.post_shift = post_affine[0].data_ptr(), post_affine[1].data_ptr()
    if pool is not None:      # fused 2x2 max-pool of the written values (eval mode)
        assert (pool.B, pool.H, pool.W, pool.C) == (x.B, Ho // 2, Wo // 2, Cout) and pool.t.dtype == x.t.dtype
        d.pool_out, d.pool_ld = pool.ptr, pool.ld
    assert wt.shape[0] == KH * KW and (virt is not None or wt.shape[2] >= x.C) and wt.dtype == x.t.dtype and out.t.dtype == x.t.dtype

    if pixel_shuffle:
        assert (out.H, out.W, out.C) == (2 * Ho, 2 * Wo, Cout // 4), "pixel-shuffled output view mismatch"
    else:
        assert (out.H, out.W, out.C) == (Ho, Wo, Cout), f"output view mismatch {(out.H, out.W, out.C)} vs {(Ho, Wo, Cout)}"
    assert out.B == x.B
    # same dispatch rule as cvcs_conv2d: 3x3 / stride 1 / pad 1 on maps of at least 8 pixels -> the halo kernel
    halo = KH == 3 and KW == 3 and stride == 1 and pad == 1 and dil == 1 and not pixel_shuffle and x.H >= 8 and x.W >= 8
    taps = (not halo and x.code == BF16 and virt is None and pad == 0 and dil == 1 and Cout % 128 == 0 and
            not (stats is not None and relu) and post_affine is None and
            (pre_affine is None or ((KH, KW, stride) == (1, 1, 1) and stats is None and not pixel_shuffle)) and
            ((KH, KW, stride) == (1, 1, 1) or (KH, KW, stride) == (2, 2, 2) or
             ((KH, KW, stride) == (1, 1, 2) and x.H % 2 == 0 and x.W % 2 == 0 and pre_affine is None and not pixel_shuffle)))
    # algorithmic FLOPs: zero-padded input channels (the first layer's 3 -> one K-group) do not count; `flops` overrides
    # (a zero-dilated strided data gradient multiplies four times the pixels its convolution has).
    # launches that also carry the first pass of a BatchNorm backward are timed as their own family: their duration is not
    # that of a convolution alone
    fam = ("conv3x3_halo_bn_bwd" if bn_bwd is not None else "conv3x3_halo") if halo else ("conv_taps" if taps else "conv_igemm")
    if SCOPE:
        fam = f"{fam}:{SCOPE}"
    # (a second contraction source carries the BatchNorm-backward correction of the Gram path: extra work of that formulation, not algorithmic)
    tag = (fam, float(flops) if flops is not None else 2.0 * x.B * Ho * Wo * Cout * (cin_real or (x.C if virt is None else 21)) * KH * KW)
    if _lib._recording is not None:
        _lib.pending_tag = tag
    elif TIMERS is not None:
        ev = TIMERS.bracket(*tag)
        ev[0].record()
    check(_lib.lib().cvcs_conv2d(C.byref(d), _stream()), "cvcs_conv2d")
    if TIMERS is not None and _lib._recording is None:
        ev[1].record()


def wgrad_workspace_floats(B, Ho, Wo, Cout, Cin, KH, KW, stride) -> int:
    n = _lib.lib().cvcs_wgrad_slices(B, Ho, Wo, Cout, Cin, KH, KW, stride)
    if n < 0:
        raise _lib.CvcsError("cvcs_wgrad_slices: bad shape")
    return n * KH * KW * Cout * Cin


def _wgrad_desc(x: View, dy: View, KH, KW, stride, pad, virt=None, dil=1):
    d = WgradDesc()
    d.dil = dil
    if virt is not None:   # the stem's weight gradient over virtual pixels (see _conv_desc)
        assert (KH, KW, stride, pad) == (7, 1, 2, 3) and x.C == 4 and x.off == 0
        Wv = (x.W - 8) // 2
        d.x, d.x_ld, d.B, d.H, d.W, d.Cin = x.ptr, 8, x.B, x.H, Wv, 32
        d.aniso, d.stride_w, d.pad_w = 1, 1, 0
        d.x_row_pitch, d.x_img_pitch = x.W * 4, x.H * x.W * 4
    else:
        d.x, d.x_ld, d.B, d.H, d.W, d.Cin = x.ptr, x.ld, x.B, x.H, x.W, x.C
    if isinstance(x, UpView):
        d.x_up2 = 1
    d.dy, d.dy_ld, d.Ho, d.Wo, d.Cout = dy.ptr, dy.ld, dy.H, dy.W, dy.C
    d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
    d.Cin_real = d.Cin         # (conv2d_wgrad overrides it for the zero-padded first layer; the workspace query sees the same value)
    d.dtype = x.code
    return d


def wgrad_takes_bias(x: View, dy: View, KH, KW, stride, pad, dil=1) -> bool:
    return bool(_lib.lib().cvcs_wgrad_takes_bias(C.byref(_wgrad_desc(x, dy, KH, KW, stride, pad, None, dil))))


def wgrad_workspace_floats_for(x: View, dy: View, KH, KW, stride, pad, virt=None, dil=1, with_bias=False) -> int:
    """exact split-K workspace (floats) of one weight-gradient launch (with_bias: the launch also produces the bias gradient)"""
    d = _wgrad_desc(x, dy, KH, KW, stride, pad, virt, dil)
    if with_bias:
        d.dbias = 16      # (any non-null value: only its presence matters to the size query)
    n = _lib.lib().cvcs_wgrad_workspace_floats(C.byref(d))
    if n < 0:
        raise _lib.CvcsError("cvcs_wgrad_workspace_floats: bad shape")
    return n


def conv2d_wgrad(x: View, dy: View, dw: torch.Tensor, KH, KW, stride, pad, workspace: torch.Tensor, cin_real=None, virt=None,
                 dil=1, dbias: torch.Tensor | None = None):
    """dw[Cout][Cin_real][KH][KW] (f32, contiguous) = sum_p dy[p] (x) x[pix(p, tap)]; dbias (optional, f32 [Cout]) = sum_p dy[p]
    out of the same launch where wgrad_takes_bias() says so."""
    d = _wgrad_desc(x, dy, KH, KW, stride, pad, virt, dil)
    cin_real = d.Cin if cin_real is None else cin_real
    d.dw, d.Cin_real = dw.data_ptr(), cin_real
    if dbias is not None:
        assert dbias.dtype == torch.float32 and dbias.is_contiguous() and dbias.numel() == dy.C
        d.dbias = dbias.data_ptr()
    d.workspace = workspace.data_ptr()
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == dy.C * cin_real * KH * KW
    assert workspace.dtype == torch.float32 and \
        workspace.numel() >= _lib.lib().cvcs_wgrad_workspace_floats(C.byref(d)), "wgrad workspace too small"
    tag = ("wgrad", 2.0 * x.B * dy.H * dy.W * dy.C * (21 if virt is not None else cin_real) * KH * KW)
    if _lib._recording is not None:
        _lib.pending_tag = tag
    elif TIMERS is not None:
        ev = TIMERS.bracket(*tag)
        ev[0].record()
    check(_lib.lib().cvcs_conv2d_wgrad(C.byref(d), _stream()), "cvcs_conv2d_wgrad")
    if TIMERS is not None and _lib._recording is None:
        ev[1].record()


# ------------------------------------------------------------------------------------------------ fp8 convolutions
FP8_MAX = {_lib.E4M3: 448.0, _lib.E5M2: 57344.0}
FP8_TORCH = {_lib.E4M3: torch.float8_e4m3fn, _lib.E5M2: torch.float8_e5m2}
FP8_MARGIN = 2.0      # delayed scaling: scale = FP8_MARGIN * amax(previous step) / fmax - one binade of headroom for the next step


class Fp8Slots:
    """per-tensor scale slots of the fp8 path: one device table [n][SLOT] f32, a slot = (-, scale, 1 / scale, fmax, amax the scale came from)
    on its first 64-byte line + 64 partial amax words on lines of their own (include/cvcs_hip.h).  The kernels read scale / 1/scale through
    pointers into it; cvcs_fp8_update_scales turns the step's amax into the next step's scale."""
    SLOT = 1040

    def __init__(self, device, capacity=256):
        self.t = torch.zeros((capacity, self.SLOT), dtype=torch.float32, device=device)
        self.t[:, 1:3] = 1.0
        self.names, self.fresh = {}, set()

    def slot(self, name, fmt):
        """index of the slot called `name` (created on first use: scale 1, no history)"""
        i = self.names.get(name)
        if i is None:
            i = self.names[name] = len(self.names)
            assert i < self.t.shape[0], "Fp8Slots capacity"
            self.t[i, 3] = FP8_MAX[fmt]
            self.fresh.add(i)
        return i

    def ptr(self, i, field=0):
        return self.t.data_ptr() + (self.SLOT * i + field) * 4

    def amax_now(self, i) -> float:
        """the amax collected so far in this step (maximum over the partial words)"""
        return float(self.t[i, 16::16].max())

    def scale(self, i) -> float:
        return float(self.t[i, 1])

    def update(self):
        check(_lib.lib().cvcs_fp8_update_scales(self.t.data_ptr(), max(len(self.names), 1), FP8_MARGIN, _stream()), "cvcs_fp8_update_scales")


def quantize_fp8(x: View, q: View | None, fmt: int, slots: Fp8Slots, i: int, take_amax=True):
    """q = fmt(x / scale_i) (bf16 view -> fp8 view) and amax_i = max(amax_i, max|x|).  A slot without history is CALIBRATED first, outside
    any recording: the amax of this very tensor becomes its scale (current scaling for a slot's first use, delayed scaling afterwards)"""
    M = x.B * x.H * x.W
    assert x.code == BF16 and x.C % 16 == 0
    if i in slots.fresh:
        h = _lib._load()      # (the raw handle: calibration launches are not part of a recorded plan)
        check(h.cvcs_quantize_fp8(x.ptr, x.ld, M, x.C, 0, 0, fmt, slots.ptr(i), 1, _stream()), "cvcs_quantize_fp8(amax)")
        check(h.cvcs_fp8_update_scales(slots.ptr(i), 1, FP8_MARGIN, _stream()), "cvcs_fp8_update_scales")
        slots.fresh.discard(i)
    if q is None:
        return
    assert q.t.dtype == torch.uint8 and (q.B, q.H, q.W, q.C) == (x.B, x.H, x.W, x.C)
    _tag_hbm("quantize_fp8", M * x.C * 3)
    check(_lib.lib().cvcs_quantize_fp8(x.ptr, x.ld, M, x.C, q.ptr, q.ld, fmt, slots.ptr(i), int(take_amax), _stream()), "cvcs_quantize_fp8")


def conv3x3_fp8(x8: View, fmt: int, wt8: torch.Tensor, out: View, slots: Fp8Slots, ix: int, iw: int, stats=None, relu=False, pre_affine=None):
    """out (bf16) = act(s_x s_w conv3x3(x8, wt8)) on the block-scaled fp8 MFMA; wt8 u8 [9][Cout][Cin] (e4m3); x8 u8 view"""
    Cout, Cin = wt8.shape[1], wt8.shape[2]
    assert x8.t.dtype == torch.uint8 and wt8.dtype == torch.uint8 and wt8.shape[0] == 9 and x8.C == Cin and out.code == BF16
    assert (out.B, out.H, out.W, out.C) == (x8.B, x8.H, x8.W, Cout)
    d = _lib.Conv8Desc()
    d.in_, d.in_ld, d.B, d.H, d.W, d.Cin, d.in_fmt = x8.ptr, x8.ld, x8.B, x8.H, x8.W, Cin, fmt
    d.wt, d.out, d.out_ld, d.Cout = wt8.data_ptr(), out.ptr, out.ld, Cout
    d.scale_in, d.scale_w = slots.ptr(ix, 1), slots.ptr(iw, 1)
    d.relu = int(relu)
    if pre_affine is not None:
        d.pre_scale, d.pre_shift = pre_affine[0].data_ptr(), pre_affine[1].data_ptr()
    if stats is not None:
        d.stat_sum, d.stat_m2, d.stat_cnt = (_ptr(t) for t in stats)
    fam = "conv3x3_fp8" + (f":{SCOPE}" if SCOPE else "")
    tag = (fam, 2.0 * x8.B * x8.H * x8.W * Cout * Cin * 9)
    if _lib._recording is not None:
        _lib.pending_tag = tag
    elif TIMERS is not None:
        ev = TIMERS.bracket(*tag)
        ev[0].record()
    check(_lib.lib().cvcs_conv3x3_fp8(C.byref(d), _stream()), "cvcs_conv3x3_fp8")
    if TIMERS is not None and _lib._recording is None:
        ev[1].record()


def fp8_stat_rows(x: View) -> int:
    return x.B * -(-x.H // 16) * -(-x.W // 16)


def _tag_hbm(family, nbytes):
    """timer family of an HBM-bound pass inside a recorded plan: `work` is its ALGORITHMIC byte count (every input read once, every
    output written once at storage precision) - bench.py prices these families against the 8 TB/s HBM peak"""
    if _lib._recording is not None:
        _lib.pending_tag = ("hbm_" + family + (f":{SCOPE}" if SCOPE else ""), float(nbytes))


# ------------------------------------------------------------------------------------------------ batch norm
def bn_finalize_workspace_floats(rows: int, C_: int) -> int:
    return _lib.lib().cvcs_bn_finalize_workspace_floats(rows, C_)


def bn_finalize(stats, rows, M, C_, gamma, beta, rmean, rvar, train, scale, shift, save_mean, save_invstd,
                momentum=0.1, eps=1e-5, workspace=None):
    """stats = (sum, m2, cnt) as written by conv2d (ignored when train is False)"""
    ssum, sm2, scnt = stats if stats is not None else (None, None, None)
    if train and workspace is None and bn_finalize_workspace_floats(rows, C_) > 0:
        workspace = torch.empty(bn_finalize_workspace_floats(rows, C_), dtype=torch.float32, device=scale.device)
    check(_lib.lib().cvcs_bn_finalize(_ptr(ssum), _ptr(sm2), _ptr(scnt), rows, M, C_, gamma.data_ptr(), beta.data_ptr(),
                                      rmean.data_ptr(), rvar.data_ptr(), momentum, eps, int(train), scale.data_ptr(),
                                      shift.data_ptr(), _ptr(save_mean), _ptr(save_invstd), _ptr(workspace), _stream()),
          "cvcs_bn_finalize")


def bn_moments(stats, rows, C_, moments, workspace=None):
    """this rank's per-channel (n, sum x, sum x^2) as f64 [3*C] from the conv's partial statistics"""
    ssum, sm2, scnt = stats
    assert moments.dtype == torch.float64 and moments.numel() >= 3 * C_
    if workspace is None and bn_finalize_workspace_floats(rows, C_) > 0:
        workspace = torch.empty(bn_finalize_workspace_floats(rows, C_), dtype=torch.float32, device=moments.device)
    check(_lib.lib().cvcs_bn_moments(ssum.data_ptr(), sm2.data_ptr(), scnt.data_ptr(), rows, C_, moments.data_ptr(),
                                     _ptr(workspace), _stream()), "cvcs_bn_moments")


def bn_finalize_moments(moments, C_, gamma, beta, rmean, rvar, scale, shift, save_mean, save_invstd, momentum=0.1, eps=1e-5):
    check(_lib.lib().cvcs_bn_finalize_moments(moments.data_ptr(), C_, gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(),
                                              rvar.data_ptr(), momentum, eps, scale.data_ptr(), shift.data_ptr(),
                                              save_mean.data_ptr(), save_invstd.data_ptr(), _stream()),
          "cvcs_bn_finalize_moments")


def bn_bwd_coeffs(sums, M, C_, ca, cb):
    assert sums.dtype == torch.float32 and sums.numel() >= 2 * C_
    check(_lib.lib().cvcs_bn_bwd_coeffs(sums.data_ptr(), M, C_, ca.data_ptr(), cb.data_ptr(), _stream()), "cvcs_bn_bwd_coeffs")


def _q8_args(q8):
    """q8 = (fp8 view, format, Fp8Slots, slot index, take_amax) -> the trailing C arguments of a *_q8 producer"""
    qv, fmt, slots, i, take = q8
    assert qv.t.dtype == torch.uint8
    return qv.ptr, qv.ld, fmt, slots.ptr(i), int(take)


def bn_act(y: View, scale, shift, relu: bool, out: View | None, pool: View | None = None, q8=None):
    """out = relu?(scale*y + shift) (+ pool = 2x2 max of it); out=None with a pool view: pooling only.
    q8 = (fp8 view, format, slots, slot, take_amax): also the fp8 image of out (cvcs_bn_act_q8)"""
    _tag_hbm("bn_act", y.B * y.H * y.W * y.C * (y.t.element_size() * (2 if pool is None else 2.25) + (1 if q8 is not None else 0)))
    if q8 is not None:
        assert pool is None and out is not None and (q8[0].B, q8[0].H, q8[0].W, q8[0].C) == (y.B, y.H, y.W, y.C)
        check(_lib.lib().cvcs_bn_act_q8(y.ptr, y.ld, y.B, y.H, y.W, y.C, scale.data_ptr(), shift.data_ptr(), int(relu), out.ptr, out.ld,
                                        *_q8_args(q8), y.code, _stream()), "cvcs_bn_act_q8")
        return
    check(_lib.lib().cvcs_bn_act(y.ptr, y.ld, y.B, y.H, y.W, y.C, scale.data_ptr(), shift.data_ptr(), int(relu),
                                 0 if out is None else out.ptr, 0 if out is None else out.ld,
                                 0 if pool is None else pool.ptr, 0 if pool is None else pool.ld, y.code,
                                 _stream()), "cvcs_bn_act")


def bn_bwd_rows(M: int) -> int:
    return _lib.lib().cvcs_bn_bwd_rows(M)


def bn_bwd_reduce(y: View, g1: View, g2: View | None, scale, shift, mean, invstd, mode, part_dz, part_dzx):
    assert (g1.B, g1.H, g1.W, g1.C) == (y.B, y.H, y.W, y.C), "gradient / activation shape mismatch"
    _tag_hbm("bn_bwd", y.B * y.H * y.W * y.C * y.t.element_size() * 2)
    check(_lib.lib().cvcs_bn_bwd_reduce(y.ptr, y.ld, g1.ptr, g1.ld, 0 if g2 is None else g2.ptr, 0 if g2 is None else g2.ld,
                                        y.B, y.H, y.W, y.C, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                        invstd.data_ptr(), mode, part_dz.data_ptr(), part_dzx.data_ptr(), y.code, _stream()),
          "cvcs_bn_bwd_reduce")


def bn_bwd_finalize(part_dz, part_dzx, rows, M, C_, gamma, invstd, dgamma, dbeta, ca, cb):
    check(_lib.lib().cvcs_bn_bwd_finalize(part_dz.data_ptr(), part_dzx.data_ptr(), rows, M, C_, gamma.data_ptr(),
                                          invstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ca.data_ptr(),
                                          cb.data_ptr(), _stream()), "cvcs_bn_bwd_finalize")


def bn_bwd_apply(y: View, g1: View, g2: View | None, scale, shift, mean, invstd, ca, cb, mode, dy: View, part_db, q8=None):
    assert (g1.B, g1.H, g1.W, g1.C) == (y.B, y.H, y.W, y.C) == (dy.B, dy.H, dy.W, dy.C), "gradient / activation shape mismatch"
    _tag_hbm("bn_bwd", y.B * y.H * y.W * y.C * (y.t.element_size() * 3 + (1 if q8 is not None else 0)))
    if q8 is not None:      # also the fp8 image of dy (cvcs_bn_bwd_apply_q8)
        assert g2 is None
        check(_lib.lib().cvcs_bn_bwd_apply_q8(y.ptr, y.ld, g1.ptr, g1.ld, y.B, y.H, y.W, y.C, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                              invstd.data_ptr(), ca.data_ptr(), cb.data_ptr(), mode, dy.ptr, dy.ld, part_db.data_ptr(),
                                              *_q8_args(q8), y.code, _stream()), "cvcs_bn_bwd_apply_q8")
        return
    check(_lib.lib().cvcs_bn_bwd_apply(y.ptr, y.ld, g1.ptr, g1.ld, 0 if g2 is None else g2.ptr, 0 if g2 is None else g2.ld,
                                       y.B, y.H, y.W, y.C, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                       invstd.data_ptr(), ca.data_ptr(), cb.data_ptr(), mode, dy.ptr, dy.ld,
                                       part_db.data_ptr(), y.code, _stream()), "cvcs_bn_bwd_apply")


def colsum_finalize(part, rows, C_, out):
    check(_lib.lib().cvcs_colsum_finalize(part.data_ptr(), rows, C_, out.data_ptr(), _stream()), "cvcs_colsum_finalize")


def colsum_partial(x: View, part):
    check(_lib.lib().cvcs_colsum_partial(x.ptr, x.ld, x.B * x.H * x.W, x.C, part.data_ptr(), x.code, _stream()),
          "cvcs_colsum_partial")


# ------------------------------------------------------------------------------------------------ BatchNorm from the Gram matrix
def gram_workspace_floats(M: int, C_: int) -> int:
    n = _lib.lib().cvcs_gram_workspace_floats(M, C_)
    if n < 0:
        raise _lib.CvcsError("cvcs_gram_workspace_floats: bad shape")
    return n


def gram(a: View, G: torch.Tensor, s: torch.Tensor, workspace: torch.Tensor):
    """G (f64 [C, C]) = sum_p a_p a_p^T, s (f64 [C]) = sum_p a_p over the pixels of the bf16 view a"""
    M = a.B * a.H * a.W
    assert a.code == BF16 and G.dtype == torch.float64 and s.dtype == torch.float64 and G.numel() == a.C * a.C and s.numel() == a.C
    assert workspace.dtype == torch.float32 and workspace.numel() >= gram_workspace_floats(M, a.C)
    _tag_hbm("gram", M * a.C * 2)
    check(_lib.lib().cvcs_gram(a.ptr, a.ld, M, a.C, G.data_ptr(), s.data_ptr(), workspace.data_ptr(), _stream()), "cvcs_gram")


def bn_gram_finalize(G, s, w, M, gamma, beta, rmean, rvar, scale, shift, save_mean, save_invstd, q, momentum=0.1, eps=1e-5):
    """train-mode BatchNorm statistics of y = w a from the Gram matrix of a (w: bf16 [C, m], the conv's packed forward weight); q <- w G"""
    C_, m = w.shape[-2], w.shape[-1]
    assert w.dtype == torch.bfloat16 and w.is_contiguous() and q.dtype == torch.float32 and q.numel() == C_ * m and G.numel() == m * m
    check(_lib.lib().cvcs_bn_gram_finalize(G.data_ptr(), s.data_ptr(), w.data_ptr(), C_, m, M, gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(),
                                           rvar.data_ptr(), momentum, eps, scale.data_ptr(), shift.data_ptr(), save_mean.data_ptr(),
                                           save_invstd.data_ptr(), q.data_ptr(), _stream()), "cvcs_bn_gram_finalize")


def bn_gram_bwd(r, sum_dz, w, q, s, M, scale, save_mean, save_invstd, dgamma, dbeta, dw, wd2, coef):
    """BatchNorm backward of y = w a without y: r = dz^T a (f32 [C, m]), sum_dz (f32 [C]) -> dgamma, dbeta, dw (f32 [C, m]), the dz columns of
    the packed two-source data-gradient weight wd2 (bf16 [m, C + m]), coef (f32 [2, C]: kappa, beta0)"""
    C_, m = w.shape[-2], w.shape[-1]
    assert r.dtype == torch.float32 and r.numel() == C_ * m and dw.numel() == C_ * m and dw.is_contiguous() and wd2.dtype == torch.bfloat16
    assert wd2.numel() == m * (C_ + m) and coef.numel() == 2 * C_ and sum_dz.numel() == C_
    check(_lib.lib().cvcs_bn_gram_bwd(r.data_ptr(), sum_dz.data_ptr(), w.data_ptr(), q.data_ptr(), s.data_ptr(), C_, m, M, scale.data_ptr(),
                                      save_mean.data_ptr(), save_invstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), dw.data_ptr(),
                                      wd2.data_ptr(), coef.data_ptr(), _stream()), "cvcs_bn_gram_bwd")


def bn_gram_mmat_workspace_floats(C_: int, m: int) -> int:
    return _lib.lib().cvcs_bn_gram_mmat_workspace_floats(C_, m)


def bn_gram_mmat(w, coef, wd2, bias, workspace=None):
    """the `a` columns of wd2 (w^T diag(kappa) w) and the data gradient's bias w^T beta0 (f32 [m])"""
    C_, m = w.shape[-2], w.shape[-1]
    assert bias.dtype == torch.float32 and bias.numel() == m
    if workspace is None:
        workspace = torch.empty(bn_gram_mmat_workspace_floats(C_, m), dtype=torch.float32, device=w.device)
    assert workspace.dtype == torch.float32 and workspace.numel() >= bn_gram_mmat_workspace_floats(C_, m)
    check(_lib.lib().cvcs_bn_gram_mmat(w.data_ptr(), coef.data_ptr(), C_, m, wd2.data_ptr(), bias.data_ptr(), workspace.data_ptr(), _stream()),
          "cvcs_bn_gram_mmat")


def bn_gram_fold(w1, scale1, shift1, w2, scale2, shift2, w_out, bias):
    """w_out [C][m1 + m2] = [scale1 * w1 | scale2 * w2] (bf16), bias = shift1 + shift2: the operand of the two-source GEMM that is a bottleneck
    tail with a same-resolution projection shortcut (both BatchNorms from Gram matrices)"""
    C_, m1, m2 = w1.shape[-2], w1.shape[-1], w2.shape[-1]
    assert w2.shape[-2] == C_ and w_out.numel() == C_ * (m1 + m2) and bias.dtype == torch.float32 and bias.numel() == C_
    assert w1.dtype == w2.dtype == w_out.dtype == torch.bfloat16 and w1.is_contiguous() and w2.is_contiguous() and w_out.is_contiguous()
    check(_lib.lib().cvcs_bn_gram_fold(w1.data_ptr(), scale1.data_ptr(), shift1.data_ptr(), m1, w2.data_ptr(), scale2.data_ptr(), shift2.data_ptr(), m2,
                                       C_, w_out.data_ptr(), bias.data_ptr(), _stream()), "cvcs_bn_gram_fold")


# ------------------------------------------------------------------------------------------------ residual networks
def bn_add_act(y1: View, s1, b1, y2: View, s2, b2, out: View, q8=None):
    """out = relu(s1*y1 + b1 + (s2*y2 + b2 | y2)): the tail of a residual block in one pass (q8: also its fp8 image, cvcs_bn_add_act_q8)"""
    M = y1.B * y1.H * y1.W
    assert (y2.B, y2.H, y2.W, y2.C) == (y1.B, y1.H, y1.W, y1.C) == (out.B, out.H, out.W, out.C)
    _tag_hbm("residual", M * y1.C * (y1.t.element_size() * 3 + (1 if q8 is not None else 0)))
    if q8 is not None:
        check(_lib.lib().cvcs_bn_add_act_q8(y1.ptr, y1.ld, s1.data_ptr(), b1.data_ptr(), y2.ptr, y2.ld, _ptr(s2), _ptr(b2), M, y1.C, out.ptr, out.ld,
                                            *_q8_args(q8), y1.code, _stream()), "cvcs_bn_add_act_q8")
        return
    check(_lib.lib().cvcs_bn_add_act(y1.ptr, y1.ld, s1.data_ptr(), b1.data_ptr(), y2.ptr, y2.ld, _ptr(s2), _ptr(b2), M, y1.C,
                                     out.ptr, out.ld, y1.code, _stream()), "cvcs_bn_add_act")


def relu_bwd_sum_bn(out: View | None, grads, dz: View, bns, part_dz, part_dzx, pooled=None):
    """relu_bwd_sum + the reduce pass of up to two BatchNorm backwards that consume dz: bns = [(y view, mean, invstd)], part_dzx = one buffer
    per BatchNorm; partial rows = bn_bwd_rows(B*H*W).  pooled = ([gradient views of the MaxPool2d(3, 2, 1) of this map], idx): the max-pool
    backward as one more term of the sum (the ResNet stem)"""
    assert (1 <= len(grads) <= 3 or (pooled is not None and len(grads) <= 3)) and 1 <= len(bns) <= 2 and len(part_dzx) == len(bns)
    d = _lib.TailBwdDesc()
    if pooled is not None:
        pg, idx = pooled
        assert 1 <= len(pg) <= 2 and idx.dtype == torch.uint8
        for j, v in enumerate(pg):
            assert (v.B, v.H, v.W, v.C) == (dz.B, (dz.H + 1) // 2, (dz.W + 1) // 2, dz.C) and v.t.dtype == dz.t.dtype, "pooled gradient view mismatch"
            d.pool_g[j], d.pool_g_ld[j] = v.ptr, v.ld
        assert idx.numel() == pg[0].B * pg[0].H * pg[0].W * pg[0].C
        d.pool_idx = idx.data_ptr()
    if out is not None:
        d.out, d.out_ld = out.ptr, out.ld
    nbytes = dz.B * dz.H * dz.W * dz.C * dz.t.element_size()
    units = 1.0 + (out is not None) + len(bns) + (0.375 * len(pooled[0]) if pooled is not None else 0.0)
    for j, (v, half) in enumerate(grads):
        want = (dz.B, dz.H // 2, dz.W // 2, dz.C) if half else (dz.B, dz.H, dz.W, dz.C)
        assert (v.B, v.H, v.W, v.C) == want and v.t.dtype == dz.t.dtype, "gradient view mismatch"
        d.g[j], d.g_ld[j], d.g_half[j] = v.ptr, v.ld, int(half)
        units += 0.25 if half else 1.0
    d.dz, d.dz_ld, d.B, d.H, d.W, d.C, d.dtype = dz.ptr, dz.ld, dz.B, dz.H, dz.W, dz.C, dz.code
    rows = bn_bwd_rows(dz.B * dz.H * dz.W)
    assert part_dz.numel() >= rows * dz.C and all(p.numel() >= rows * dz.C for p in part_dzx)
    d.part_dz = part_dz.data_ptr()
    for j, ((yv, mean, invstd), pz) in enumerate(zip(bns, part_dzx)):
        assert (yv.B, yv.H, yv.W, yv.C) == (dz.B, dz.H, dz.W, dz.C) and yv.t.dtype == dz.t.dtype
        d.y[j], d.y_ld[j], d.mean[j], d.invstd[j], d.part_dzx[j] = yv.ptr, yv.ld, mean.data_ptr(), invstd.data_ptr(), pz.data_ptr()
    _tag_hbm("residual", nbytes * units)
    check(_lib.lib().cvcs_relu_bwd_sum_bn(C.byref(d), _stream()), "cvcs_relu_bwd_sum_bn")
    return rows


def relu_bwd_sum(out: View | None, grads, dz: View):
    """dz = (sum of up to three gradients) * (out > 0); grads = [(view, half_resolution: bool)]"""
    assert 1 <= len(grads) <= 3
    g = list(grads) + [(None, False)] * (3 - len(grads))
    for v, half in grads:
        want = (dz.B, dz.H // 2, dz.W // 2, dz.C) if half else (dz.B, dz.H, dz.W, dz.C)
        assert (v.B, v.H, v.W, v.C) == want and v.t.dtype == dz.t.dtype, "gradient view mismatch"
    a = []
    for v, half in g:
        a += [0 if v is None else v.ptr, 0 if v is None else v.ld, int(half)]
    _tag_hbm("residual", dz.B * dz.H * dz.W * dz.C * dz.t.element_size() * (1 + (out is not None) + sum(0.25 if half else 1.0 for _, half in grads)))
    check(_lib.lib().cvcs_relu_bwd_sum(0 if out is None else out.ptr, 0 if out is None else out.ld, *a, dz.B, dz.H, dz.W, dz.C,
                                       dz.ptr, dz.ld, dz.code, _stream()), "cvcs_relu_bwd_sum")


def maxpool3x3s2_fwd(x: View, out: View, idx: torch.Tensor):
    assert idx.dtype == torch.uint8 and idx.is_contiguous() and idx.numel() == out.B * out.H * out.W * out.C
    assert (out.H, out.W, out.C) == ((x.H + 1) // 2, (x.W + 1) // 2, x.C)
    check(_lib.lib().cvcs_maxpool3x3s2_fwd(x.ptr, x.ld, x.B, x.H, x.W, x.C, out.ptr, out.ld, idx.data_ptr(), x.code, _stream()),
          "cvcs_maxpool3x3s2_fwd")


def maxpool3x3s2_bwd(g0: View, g1: View | None, idx: torch.Tensor, dx: View):
    assert (g0.H, g0.W, g0.C) == ((dx.H + 1) // 2, (dx.W + 1) // 2, dx.C) and idx.numel() == g0.B * g0.H * g0.W * g0.C
    check(_lib.lib().cvcs_maxpool3x3s2_bwd(g0.ptr, g0.ld, 0 if g1 is None else g1.ptr, 0 if g1 is None else g1.ld, idx.data_ptr(),
                                           dx.B, dx.H, dx.W, dx.C, dx.ptr, dx.ld, dx.code, _stream()), "cvcs_maxpool3x3s2_bwd")


def dilate2x(x: View, out: View):
    assert (out.B, out.H, out.W, out.C) == (x.B, 2 * x.H, 2 * x.W, x.C)
    check(_lib.lib().cvcs_dilate2x(x.ptr, x.ld, x.B, x.H, x.W, x.C, out.ptr, out.ld, x.code, _stream()), "cvcs_dilate2x")


def regrid(x: View, out: View, H, W, a, b):
    """x: the [B, H, W, C] map in sub-grid layout L_a ([B*a*a, H/a, W/a, C]) -> out in L_b (cvcs_regrid)"""
    B = x.B // (a * a)
    assert (x.B, x.H, x.W) == (B * a * a, H // a, W // a) and (out.B, out.H, out.W, out.C) == (B * b * b, H // b, W // b, x.C)
    check(_lib.lib().cvcs_regrid(x.ptr, x.ld, B, H, W, x.C, a, b, out.ptr, out.ld, x.code, _stream()), "cvcs_regrid")


def pack_input_stem(src: torch.Tensor, dst: torch.Tensor):
    """src NCHW u8|f32 [B,3,H,W] -> dst [B,H,W+8,4] (real column x at x+3; margins and channel 3 zero)"""
    assert src.is_contiguous() and dst.is_contiguous() and src.dtype in (torch.uint8, torch.float32)
    B, C_, H, W = src.shape
    assert C_ == 3 and tuple(dst.shape) == (B, H, W + 8, 4)
    check(_lib.lib().cvcs_pack_input_stem(src.data_ptr(), int(src.dtype == torch.uint8), B, H, W, dst.data_ptr(),
                                          dtype_code(dst.dtype), _stream()), "cvcs_pack_input_stem")


def pack_stem_weight(w: torch.Tensor, wf: torch.Tensor):
    assert w.dtype == torch.float32 and w.is_contiguous() and tuple(w.shape[1:]) == (3, 7, 7) and tuple(wf.shape) == (7, w.shape[0], 32)
    check(_lib.lib().cvcs_pack_stem_weight(w.data_ptr(), w.shape[0], wf.data_ptr(), dtype_code(wf.dtype), _stream()),
          "cvcs_pack_stem_weight")


def unpack_stem_wgrad(tmp: torch.Tensor, dw: torch.Tensor):
    assert tmp.dtype == torch.float32 and dw.dtype == torch.float32 and dw.is_contiguous() and tmp.numel() == dw.shape[0] * 32 * 7
    check(_lib.lib().cvcs_unpack_stem_wgrad(tmp.data_ptr(), dw.shape[0], dw.data_ptr(), _stream()), "cvcs_unpack_stem_wgrad")


# ------------------------------------------------------------------------------------------------ Swin-T + UPerNet
def layernorm_rows(M: int) -> int:
    return _lib.lib().cvcs_layernorm_rows(M)


def layernorm_fwd(x: View, gamma, beta, out: View, mean, invstd, eps=1e-5):
    M = x.B * x.H * x.W
    check(_lib.lib().cvcs_layernorm_fwd(x.ptr, x.ld, M, x.C, gamma.data_ptr(), beta.data_ptr(), eps, out.ptr, out.ld, mean.data_ptr(),
                                        invstd.data_ptr(), x.code, _stream()), "cvcs_layernorm_fwd")


def layernorm_bwd(x: View, g: View, gamma, mean, invstd, dx: View, part):
    M = x.B * x.H * x.W
    check(_lib.lib().cvcs_layernorm_bwd(x.ptr, x.ld, g.ptr, g.ld, M, x.C, gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(), dx.ptr,
                                        dx.ld, part.data_ptr(), x.code, _stream()), "cvcs_layernorm_bwd")


def gelu(x: View, out: View, g: View | None = None):
    """g None: out = gelu(x) (erf form); else out = g * gelu'(x)"""
    M = x.B * x.H * x.W
    check(_lib.lib().cvcs_gelu(x.ptr, x.ld, 0 if g is None else g.ptr, 0 if g is None else g.ld, M, x.C, out.ptr, out.ld, x.code, _stream()),
          "cvcs_gelu")


def pack_patches(src: torch.Tensor, dst: torch.Tensor):
    B, C_, H, W = src.shape
    assert C_ == 3 and src.is_contiguous() and dst.is_contiguous() and tuple(dst.shape) == (B, H // 4, W // 4, 64)
    check(_lib.lib().cvcs_pack_patches(src.data_ptr(), int(src.dtype == torch.uint8), B, H, W, dst.data_ptr(), dtype_code(dst.dtype),
                                       _stream()), "cvcs_pack_patches")


def patch_merge(fine: View, merged: View, reverse=False):
    """forward: fine [B,H,W,C] -> merged [B,H/2,W/2,4C]; reverse: merged (gradient) -> fine (gradient)"""
    src, dst = (merged, fine) if reverse else (fine, merged)
    check(_lib.lib().cvcs_patch_merge(src.ptr, src.ld, fine.B, fine.H, fine.W, fine.C, dst.ptr, dst.ld, int(reverse), fine.code, _stream()),
          "cvcs_patch_merge")


def window_gather(tokens: View, win: View, shift: int):
    check(_lib.lib().cvcs_window_gather(tokens.ptr, tokens.ld, tokens.B, tokens.H, tokens.W, tokens.C, shift, win.ptr, win.ld, tokens.code,
                                        _stream()), "cvcs_window_gather")


def window_reverse(win: View, res: View | None, out: View, shift: int):
    check(_lib.lib().cvcs_window_reverse(win.ptr, win.ld, 0 if res is None else res.ptr, 0 if res is None else res.ld, out.B, out.H, out.W,
                                         out.C, shift, out.ptr, out.ld, out.code, _stream()), "cvcs_window_reverse")


def _attn_tag(B, H, W, C_, passes):
    """timer family of the window-attention launches (replayed plans only): QK^T + AV over the padded 49-token windows = 4 * 49 * C FLOPs
    per token and pass (forward 1 pass; backward recomputes the scores and forms dP, dV, dQ, dK: 5 GEMM-shaped products = 2.5 passes)"""
    if _lib._recording is not None:
        T = B * (-(-H // 7) * 7) * (-(-W // 7) * 7)
        _lib.pending_tag = ("window_attention" + (f":{SCOPE}" if SCOPE else ""), 4.0 * 49 * C_ * T * passes)


def window_attention_fwd(qkv: View, B, H, W, heads, shift, table, out: View):
    _attn_tag(B, H, W, out.C, 1.0)
    check(_lib.lib().cvcs_window_attention_fwd(qkv.ptr, qkv.ld, B, H, W, out.C, heads, shift, table.data_ptr(), out.ptr, out.ld, qkv.code,
                                               _stream()), "cvcs_window_attention_fwd")


def window_attention_bwd_workspace(B, H, W, heads) -> int:
    return _lib.lib().cvcs_window_attention_bwd_workspace_floats(B, H, W, heads)


def window_attention_bwd(qkv: View, dout: View, B, H, W, heads, shift, table, dqkv: View, dtable, workspace):
    _attn_tag(B, H, W, dout.C, 2.5)
    check(_lib.lib().cvcs_window_attention_bwd(qkv.ptr, qkv.ld, dout.ptr, dout.ld, B, H, W, dout.C, heads, shift, table.data_ptr(), dqkv.ptr,
                                               dqkv.ld, dtable.data_ptr(), workspace.data_ptr(), qkv.code, _stream()),
          "cvcs_window_attention_bwd")


def adaptive_avg_pool(x: View, out: View, backward=False):
    """forward: x [B,H,W,C] -> out [B,S,S,C]; backward: x = gradient [B,S,S,C] -> out = gradient [B,H,W,C]"""
    big, small = (out, x) if backward else (x, out)
    check(_lib.lib().cvcs_adaptive_avg_pool(x.ptr, x.ld, big.B, big.H, big.W, big.C, small.H, out.ptr, out.ld, int(backward), x.code,
                                            _stream()), "cvcs_adaptive_avg_pool")


def resize_any(a: View, out: View, backward=False, accumulate=False):
    """forward: a [B,Hi,Wi,C] -> out [B,Ho,Wo,C] (accumulate: out +=); backward: a = gradient of the LARGE map -> out = gradient of the small"""
    small, big = (out, a) if backward else (a, out)
    check(_lib.lib().cvcs_resize_bilinear_any(a.ptr, a.ld, a.B, small.H, small.W, big.H, big.W, a.C, out.ptr, out.ld, int(backward),
                                              int(accumulate), a.code, _stream()), "cvcs_resize_bilinear_any")


# ------------------------------------------------------------------------------------------------ GroupNorm + SiLU
def gn_rows(HW: int) -> int:
    return _lib.lib().cvcs_gn_rows(HW)


def gn_stats(y: View, part):
    check(_lib.lib().cvcs_gn_stats(y.ptr, y.ld, y.B, y.H * y.W, y.C, part.data_ptr(), y.code, _stream()), "cvcs_gn_stats")


def gn_finalize(part, B, HW, C_, G, gamma, beta, scale, shift, mean, invstd, eps=1e-5):
    check(_lib.lib().cvcs_gn_finalize(part.data_ptr(), B, HW, C_, G, gamma.data_ptr(), beta.data_ptr(), eps, scale.data_ptr(),
                                      shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _stream()), "cvcs_gn_finalize")


def gn_act_fwd(y: View, scale, shift, silu: bool, out: View):
    check(_lib.lib().cvcs_gn_act_fwd(y.ptr, y.ld, y.B, y.H * y.W, y.C, scale.data_ptr(), shift.data_ptr(), int(silu), out.ptr, out.ld,
                                     y.code, _stream()), "cvcs_gn_act_fwd")


def gn_act_bwd_reduce(y: View, g: View, scale, shift, silu: bool, part):
    check(_lib.lib().cvcs_gn_act_bwd_reduce(y.ptr, y.ld, g.ptr, g.ld, y.B, y.H * y.W, y.C, scale.data_ptr(), shift.data_ptr(), int(silu),
                                            part.data_ptr(), y.code, _stream()), "cvcs_gn_act_bwd_reduce")


def gn_bwd_finalize(part, B, HW, C_, G, gamma, mean, invstd, tmp, dgamma, dbeta, ca, cb, cc):
    check(_lib.lib().cvcs_gn_bwd_finalize(part.data_ptr(), B, HW, C_, G, gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                          tmp.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ca.data_ptr(), cb.data_ptr(),
                                          cc.data_ptr(), _stream()), "cvcs_gn_bwd_finalize")


def gn_act_bwd_apply(y: View, g: View, scale, shift, silu: bool, ca, cb, cc, dy: View):
    check(_lib.lib().cvcs_gn_act_bwd_apply(y.ptr, y.ld, g.ptr, g.ld, y.B, y.H * y.W, y.C, scale.data_ptr(), shift.data_ptr(), int(silu),
                                           ca.data_ptr(), cb.data_ptr(), cc.data_ptr(), dy.ptr, dy.ld, y.code, _stream()),
          "cvcs_gn_act_bwd_apply")


# ------------------------------------------------------------------------------------------------ DeepLabV3+
def resize_bilinear_fwd(x: View, out: View, scale: int):
    assert (out.B, out.H, out.W, out.C) == (x.B, x.H * scale, x.W * scale, x.C)
    check(_lib.lib().cvcs_resize_bilinear_fwd(x.ptr, x.ld, x.B, x.H, x.W, x.C, scale, out.ptr, out.ld, x.code, _stream()),
          "cvcs_resize_bilinear_fwd")


def resize_bilinear_bwd(gout: View, gin: View, scale: int):
    assert (gout.B, gout.H, gout.W, gout.C) == (gin.B, gin.H * scale, gin.W * scale, gin.C)
    check(_lib.lib().cvcs_resize_bilinear_bwd(gout.ptr, gout.ld, gin.B, gin.H, gin.W, gin.C, scale, gin.ptr, gin.ld, gin.code,
                                              _stream()), "cvcs_resize_bilinear_bwd")


def resize_bilinear_nchw_fwd(x: torch.Tensor, out: torch.Tensor, scale: int):
    """f32 [B,NC,H,W] -> [B,NC,H*scale,W*scale]"""
    B, NC, H, W = x.shape
    assert x.dtype == torch.float32 and out.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous()
    assert tuple(out.shape) == (B, NC, H * scale, W * scale)
    check(_lib.lib().cvcs_resize_bilinear_nchw_fwd(x.data_ptr(), B * NC, H, W, scale, out.data_ptr(), _stream()),
          "cvcs_resize_bilinear_nchw_fwd")


def resize_bilinear_nchw_bwd(gout: torch.Tensor, gin: torch.Tensor, scale: int):
    B, NC, H, W = gin.shape
    assert gout.dtype == torch.float32 and gin.dtype == torch.float32 and gout.is_contiguous() and gin.is_contiguous()
    assert tuple(gout.shape) == (B, NC, H * scale, W * scale)
    check(_lib.lib().cvcs_resize_bilinear_nchw_bwd(gout.data_ptr(), B * NC, H, W, scale, gin.data_ptr(), _stream()),
          "cvcs_resize_bilinear_nchw_bwd")


def image_sum(x: View, out: View, scale: float):
    """out [B,1,1,C] = scale * per-image sum over the pixels of x"""
    assert (out.B, out.H, out.W, out.C) == (x.B, 1, 1, x.C)
    check(_lib.lib().cvcs_image_sum(x.ptr, x.ld, x.B, x.H * x.W, x.C, scale, out.ptr, out.ld, x.code, _stream()), "cvcs_image_sum")


def image_broadcast(v: View, out: View, scale: float):
    assert (v.B, v.H, v.W, v.C) == (out.B, 1, 1, out.C)
    check(_lib.lib().cvcs_image_broadcast(v.ptr, v.ld, out.B, out.H * out.W, out.C, scale, out.ptr, out.ld, out.code, _stream()),
          "cvcs_image_broadcast")


def dropout(x: View, out: View, state: torch.Tensor, p: float):
    """out = x * mask / (1 - p), mask from the counter-based hash of (state = [seed, step] int64 on the device, element index); the same call
    on a gradient is the backward"""
    assert state.dtype == torch.int64 and state.numel() == 2 and (out.B, out.H, out.W, out.C) == (x.B, x.H, x.W, x.C)
    check(_lib.lib().cvcs_dropout(x.ptr, x.ld, x.B * x.H * x.W, x.C, out.ptr, out.ld, state.data_ptr(), float(p), x.code, _stream()), "cvcs_dropout")


def counter_add(state: torch.Tensor, index: int, inc: int = 1):
    check(_lib.lib().cvcs_counter_add(state.data_ptr() + 8 * index, inc, _stream()), "cvcs_counter_add")


def dropout_mask_host(seed: int, step: int, n: int, p: float):
    """the mask cvcs_dropout applies to elements 0..n-1, replayed on the host (numpy uint64 arithmetic wraps like the device's)"""
    import numpy as np
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        z = (np.uint64(seed) ^ (np.uint64(step) * np.uint64(0xD6E8FEB86659FD93))) + idx * np.uint64(0x9E3779B97F4A7C15)
        z ^= z >> np.uint64(30); z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27); z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
        thr = np.uint64(int((np.float32(1.0) - np.float32(p)) * np.float32(16777216.0)))
        return torch.from_numpy(((z >> np.uint64(40)) < thr))


def linear_head_fwd(x: View, w, bias, logits=None, labels=None):
    """1x1 classifier on any channel count: NCHW f32 logits, or (labels given) the fused argmax"""
    NC = w.shape[0]
    assert w.dtype == torch.float32 and w.is_contiguous() and w.numel() == NC * x.C
    check(_lib.lib().cvcs_linear_head_fwd(x.ptr, x.ld, x.B, x.H, x.W, x.C, w.data_ptr(), _ptr(bias), NC, _ptr(logits), _ptr(labels),
                                          x.code, _stream()), "cvcs_linear_head_fwd")


def linear_head_bwd_rows(P: int) -> int:
    return _lib.lib().cvcs_linear_head_bwd_rows(P)


def linear_head_bwd(x: View, dlogits, w, dx: View, part_dw):
    NC = w.shape[0]
    check(_lib.lib().cvcs_linear_head_bwd(x.ptr, x.ld, dlogits.data_ptr(), x.B, x.H, x.W, x.C, w.data_ptr(), NC, dx.ptr, dx.ld,
                                          part_dw.data_ptr(), x.code, _stream()), "cvcs_linear_head_bwd")


# ------------------------------------------------------------------------------------------------ bilinear
def upsample2x_fwd(x: View, out: View, q8=None):
    _tag_hbm("resize", out.B * out.H * out.W * out.C * (out.t.element_size() * 1.25 + (1 if q8 is not None else 0)))
    if q8 is not None:
        check(_lib.lib().cvcs_upsample2x_fwd_q8(x.ptr, x.ld, x.B, x.H, x.W, x.C, out.ptr, out.ld, *_q8_args(q8), x.code, _stream()),
              "cvcs_upsample2x_fwd_q8")
        return
    check(_lib.lib().cvcs_upsample2x_fwd(x.ptr, x.ld, x.B, x.H, x.W, x.C, out.ptr, out.ld, x.code, _stream()),
          "cvcs_upsample2x_fwd")


def upsample2x_bwd(gout: View, gin: View):
    _tag_hbm("resize", gout.B * gout.H * gout.W * gout.C * gout.t.element_size() * 1.25)
    check(_lib.lib().cvcs_upsample2x_bwd(gout.ptr, gout.ld, gin.B, gin.H, gin.W, gin.C, gin.ptr, gin.ld, gin.code, _stream()),
          "cvcs_upsample2x_bwd")


# ------------------------------------------------------------------------------------------------ boundary
def pack_input(src: torch.Tensor, dst: torch.Tensor):
    """src NCHW u8|f32 [B,C,H,W] -> dst NHWC [B,H,W,Cpad] (zero padded channels)."""
    assert src.is_contiguous() and dst.is_contiguous() and src.dtype in (torch.uint8, torch.float32)
    B, C_, H, W = src.shape
    assert dst.shape[:3] == (B, H, W)
    check(_lib.lib().cvcs_pack_input(src.data_ptr(), int(src.dtype == torch.uint8), B, C_, H, W, dst.data_ptr(),
                                     dst.shape[3], dtype_code(dst.dtype), _stream()), "cvcs_pack_input")


def pack_conv_weight(w: torch.Tensor, cin_pad: int, dtype, want_dgrad=True):
    Cout, Cin, KH, KW = w.shape
    wf = torch.empty((KH * KW, Cout, cin_pad), dtype=dtype, device=w.device)
    wd = torch.empty((KH * KW, Cin, Cout), dtype=dtype, device=w.device) if want_dgrad else None
    pack_conv_weight_into(w, wf, wd)
    return wf, wd


def pack_conv_weight_into(w, wf, wd):
    Cout, Cin, KH, KW = w.shape
    assert w.is_contiguous() and w.dtype == torch.float32
    check(_lib.lib().cvcs_pack_conv_weight(w.data_ptr(), Cout, Cin, KH, KW, wf.shape[2], wf.data_ptr(), _ptr(wd),
                                           dtype_code(wf.dtype), _stream()), "cvcs_pack_conv_weight")


def pack_table(entries, device):
    """entries: [(w f32 OIHW, wf, wd | None)] -> device-resident cvcs_pack_item table (uint8 tensor) for pack_conv_weights"""
    arr = (_lib.PackItem * len(entries))()
    for it, (w, wf, wd) in zip(arr, entries):
        Cout, Cin, KH, KW = w.shape
        assert w.is_contiguous() and w.dtype == torch.float32 and Cout % 16 == 0 and KH * KW <= 9
        assert wf.shape[2] % 16 == 0 and wf.data_ptr() % 16 == 0 and (wd is None or wd.data_ptr() % 16 == 0)
        it.w, it.w_fwd, it.w_dgrad = w.data_ptr(), wf.data_ptr(), _ptr(wd)
        it.Cout, it.Cin, it.KH, it.KW, it.Cin_pad = Cout, Cin, KH, KW, wf.shape[2]
        it.Cout_pad = 0 if wd is None else wd.shape[2]      # (a data-gradient image wider than Cout must have been zero-filled by its owner)
    raw = bytes(arr)
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device), len(entries)


def pack_conv_weights(table, n, dtype):
    check(_lib.lib().cvcs_pack_conv_weights(table.data_ptr(), n, dtype_code(dtype), _stream()), "cvcs_pack_conv_weights")


def pack_convT_weight_into(w, bias, wf, bias4, wd):
    """w [Cin][Cout][2][2] -> wf [1][4*Cout][Cin], bias4 [4*Cout], wd [4][Cin][Cout]."""
    Cin, Cout = w.shape[:2]
    assert w.is_contiguous() and w.dtype == torch.float32 and tuple(w.shape[2:]) == (2, 2)
    check(_lib.lib().cvcs_pack_convT_weight(w.data_ptr(), Cin, Cout, wf.data_ptr(), bias4.data_ptr(), _ptr(bias), _ptr(wd),
                                            dtype_code(wf.dtype), _stream()), "cvcs_pack_convT_weight")


# ------------------------------------------------------------------------------------------------ head / loss
def head_fwd(x: View, w, bias, logits):
    NC = w.shape[0]
    check(_lib.lib().cvcs_head_fwd(x.ptr, x.ld, x.B, x.H, x.W, x.C, w.data_ptr(), _ptr(bias), NC, logits.data_ptr(), x.code,
                                   _stream()), "cvcs_head_fwd")


def head_fold(w, bias, scale, shift, w_out, bias_out):
    check(_lib.lib().cvcs_head_fold(w.data_ptr(), _ptr(bias), scale.data_ptr(), shift.data_ptr(), w.shape[0], w_out.data_ptr(),
                                    bias_out.data_ptr(), _stream()), "cvcs_head_fold")


def head_unfold_grad(dw, db, scale, shift):
    check(_lib.lib().cvcs_head_unfold_grad(dw.data_ptr(), db.data_ptr(), scale.data_ptr(), shift.data_ptr(), db.numel(),
                                           _stream()), "cvcs_head_unfold_grad")


def head_argmax(x: View, w, bias, labels):
    """u8 labels [B,H,W] = argmax_c (w x + b)_c (ties -> lowest class): head_fwd + argmax without the logits tensor"""
    assert labels.dtype == torch.uint8 and labels.is_contiguous() and labels.numel() == x.B * x.H * x.W
    check(_lib.lib().cvcs_head_argmax(x.ptr, x.ld, x.B, x.H, x.W, x.C, w.data_ptr(), _ptr(bias), w.shape[0], labels.data_ptr(),
                                      x.code, _stream()), "cvcs_head_argmax")


def head_bwd_rows(P: int) -> int:
    return _lib.lib().cvcs_head_bwd_rows(P)


def head_bwd(x: View, dlogits, w, dx: View, part_dw):
    NC = w.shape[0]
    check(_lib.lib().cvcs_head_bwd(x.ptr, x.ld, dlogits.data_ptr(), x.B, x.H, x.W, x.C, w.data_ptr(), NC, dx.ptr, dx.ld,
                                   part_dw.data_ptr(), x.code, _stream()), "cvcs_head_bwd")


def head_ce_rows(P: int) -> int:
    return _lib.lib().cvcs_head_ce_rows(P)


def head_ce_ok(x: View, NC: int) -> bool:
    """can cvcs_head_ce take this head?  bf16 activations with 16 / 32 / 64 channels, at most 32 classes"""
    return x.code == BF16 and x.C in (16, 32, 64) and 1 <= NC <= 32


def head_ce(x: View, w, bias, target, class_weight, ignore_index, grad_scale, dx: View, part_dw, workspace, loss_out, w_dx=None):
    """1x1 head + softmax cross-entropy + their backward in one pass over x: loss_out[0] <- mean CE, dx <- d loss / d x, part_dw <-
    head_ce_rows(P) partial rows [NC*C | NC] of dW | db.  No logits, no logit gradient."""
    NC = w.shape[0]
    P = x.B * x.H * x.W
    assert head_ce_ok(x, NC) and (dx.B, dx.H, dx.W, dx.C) == (x.B, x.H, x.W, x.C) and dx.code == BF16
    assert target.is_contiguous() and target.dtype in (torch.uint8, torch.int64) and target.numel() == P
    assert part_dw.numel() >= head_ce_rows(P) * (NC * x.C + NC) and workspace.numel() >= ce_workspace_floats(P)
    _tag_hbm("head_ce", P * (x.C * 2 * 2 + target.element_size()))
    check(_lib.lib().cvcs_head_ce(x.ptr, x.ld, x.B, x.H, x.W, x.C, w.data_ptr(), _ptr(w_dx), bias.data_ptr(), NC, target.data_ptr(),
                                  int(target.dtype == torch.uint8), _ptr(class_weight), ignore_index, grad_scale, dx.ptr, dx.ld,
                                  part_dw.data_ptr(), workspace.data_ptr(), loss_out.data_ptr(), x.code, _stream()), "cvcs_head_ce")


def ce_workspace_floats(P: int) -> int:
    return _lib.lib().cvcs_ce_workspace_floats(P)


def ce_weight_sum(target, B, NC, class_weight, ignore_index, workspace):
    """workspace[0] <- sum of class weights over this rank's non-ignored pixels (the CE denominator)"""
    HW = target.numel() // B
    assert target.is_contiguous() and target.dtype in (torch.uint8, torch.int64)
    check(_lib.lib().cvcs_ce_weight_sum(target.data_ptr(), int(target.dtype == torch.uint8), B, NC, HW, _ptr(class_weight),
                                        ignore_index, workspace.data_ptr(), _stream()), "cvcs_ce_weight_sum")


def ce_fwd_bwd(logits, target, class_weight, ignore_index, grad_scale, loss_out, dlogits, workspace,
               external_denominator=False):
    B, NC = logits.shape[:2]
    HW = logits.numel() // (B * NC)
    assert logits.is_contiguous() and logits.dtype == torch.float32 and target.is_contiguous()
    assert target.dtype in (torch.uint8, torch.int64) and target.numel() == B * HW
    check(_lib.lib().cvcs_ce_fwd_bwd(logits.data_ptr(), target.data_ptr(), int(target.dtype == torch.uint8), B, NC, HW,
                                     _ptr(class_weight), ignore_index, grad_scale, loss_out.data_ptr(), _ptr(dlogits),
                                     workspace.data_ptr(), int(external_denominator), _stream()), "cvcs_ce_fwd_bwd")


def argmax_confusion(logits, labels=None, target=None, ignore_index=-1, K=16, conf=None):
    B, NC = logits.shape[:2]
    HW = logits.numel() // (B * NC)
    assert logits.is_contiguous() and logits.dtype == torch.float32
    if target is not None:
        assert target.is_contiguous() and target.dtype in (torch.uint8, torch.int64) and target.numel() == B * HW
    check(_lib.lib().cvcs_argmax_confusion(logits.data_ptr(), B, NC, HW, _ptr(labels), _ptr(target),
                                           int(target is not None and target.dtype == torch.uint8), ignore_index, K,
                                           _ptr(conf), _stream()), "cvcs_argmax_confusion")


def label_confusion(pred, target, ignore_index, K, conf):
    assert pred.dtype == torch.uint8 and pred.is_contiguous() and target.is_contiguous() and pred.numel() == target.numel()
    assert target.dtype in (torch.uint8, torch.int64) and conf.dtype == torch.int64 and conf.numel() == K * K
    check(_lib.lib().cvcs_label_confusion(pred.data_ptr(), target.data_ptr(), int(target.dtype == torch.uint8), pred.numel(),
                                          ignore_index, K, conf.data_ptr(), _stream()), "cvcs_label_confusion")


def vote_labels(labels, out):
    """labels u8 [M, ...] -> out u8 [...]: per-pixel majority, ties -> smallest label"""
    assert labels.dtype == torch.uint8 and out.dtype == torch.uint8 and labels.is_contiguous() and out.is_contiguous()
    M = labels.shape[0]
    assert out.numel() * M == labels.numel()
    check(_lib.lib().cvcs_vote_labels(labels.data_ptr(), M, out.numel(), out.data_ptr(), _stream()), "cvcs_vote_labels")


def crop_tiles(image_chw, dst, first_tile, tiles_per_row, p):
    """image u8 [C,H,W] -> dst u8 [n,C,S,S]: S-windows centred on tiles first_tile.. of side p"""
    assert image_chw.dtype == torch.uint8 and dst.dtype == torch.uint8 and image_chw.is_contiguous() and dst.is_contiguous()
    C_, H, W = image_chw.shape
    n, _, S, _ = dst.shape
    check(_lib.lib().cvcs_crop_tiles(image_chw.data_ptr(), C_, H, W, dst.data_ptr(), n, first_tile, tiles_per_row, p, S,
                                     _stream()), "cvcs_crop_tiles")


def argmax_stitch(logits, p, first_tile, tiles_per_row, palette, rgb, labels):
    n, NC, S, _ = logits.shape
    Hout, Wout = (rgb.shape[0], rgb.shape[1]) if rgb is not None else labels.shape
    assert logits.is_contiguous() and logits.dtype == torch.float32
    check(_lib.lib().cvcs_argmax_stitch(logits.data_ptr(), n, NC, S, p, first_tile, tiles_per_row, _ptr(palette), _ptr(rgb),
                                        _ptr(labels), Hout, Wout, _stream()), "cvcs_argmax_stitch")


def label_stitch(tiles, p, first_tile, tiles_per_row, palette, rgb, labels):
    """tiles u8 [n,S,S] (per-tile label maps) -> centre p x p of each written into the full-size rgb / label maps"""
    n, S, _ = tiles.shape
    Hout, Wout = (rgb.shape[0], rgb.shape[1]) if rgb is not None else labels.shape
    assert tiles.is_contiguous() and tiles.dtype == torch.uint8
    check(_lib.lib().cvcs_label_stitch(tiles.data_ptr(), n, S, p, first_tile, tiles_per_row, _ptr(palette), _ptr(rgb),
                                       _ptr(labels), Hout, Wout, _stream()), "cvcs_label_stitch")


def gather_tiles(src, origins, dst):
    """src u8 [C,H,W] (or [H,W]) resident on the device, origins int32 [n,2] (top, left) -> dst u8 [n,C,S,S] (or [n,S,S])"""
    assert src.dtype == torch.uint8 and dst.dtype == torch.uint8 and origins.dtype == torch.int32
    assert src.is_contiguous() and dst.is_contiguous() and origins.is_contiguous()
    C_ = src.shape[0] if src.dim() == 3 else 1
    H, W = src.shape[-2:]
    n, S = dst.shape[0], dst.shape[-1]
    assert origins.shape == (n, 2)
    check(_lib.lib().cvcs_gather_tiles(src.data_ptr(), C_, H, W, origins.data_ptr(), n, S, dst.data_ptr(), _stream()),
          "cvcs_gather_tiles")


def label_histogram(labels, counts):
    assert labels.dtype == torch.uint8 and labels.is_contiguous() and counts.dtype == torch.int64
    check(_lib.lib().cvcs_label_histogram(labels.data_ptr(), labels.numel(), counts.numel(), counts.data_ptr(), _stream()),
          "cvcs_label_histogram")


# ------------------------------------------------------------------------------------------------ optimisers
def sgd_step(p, g, buf, lr, momentum, weight_decay, grad_scale, first_step):
    check(_lib.lib().cvcs_sgd_step(p.data_ptr(), g.data_ptr(), buf.data_ptr(), p.numel(), lr, momentum, weight_decay,
                                   grad_scale, int(first_step), _stream()), "cvcs_sgd_step")


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, grad_scale, step):
    check(_lib.lib().cvcs_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps,
                                    weight_decay, grad_scale, step, _stream()), "cvcs_adam_step")


# ------------------------------------------------------------------------------------------------ SegformerMod pieces (csrc/segformer.hip)
def _bytes(*views):
    return sum(v.B * v.H * v.W * v.C * v.t.element_size() for v in views)


def im2col(x: View, col: View, KH, KW, stride, oy0, ox0, direction=1):
    """col [B,Ho,Wo,Kp] <- patches of x [B,H,W,C]: col[.., (ky*KW+kx)*C + c] = x[oy*stride + oy0 + ky*direction, ox*stride + ox0 + kx*direction, c]"""
    assert col.B == x.B and col.C >= KH * KW * x.C
    _tag_hbm("patches", _bytes(col) * 2)
    check(_lib.lib().cvcs_im2col(x.ptr, x.ld, x.B, x.H, x.W, x.C, KH, KW, stride, oy0, ox0, direction, col.H, col.W, col.ptr, col.ld, x.code,
                                 _stream()), "cvcs_im2col")


SEGFORMER_MEAN, SEGFORMER_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)      # S/nets.py:339


def im2col_stem(x: torch.Tensor, col: View, KH, KW, stride, pad, mean=SEGFORMER_MEAN, std=SEGFORMER_STD):
    """the first patch embedding's patches straight from the planar u8 / f32 tile, normalised as S/nets.py:337-340"""
    import ctypes as C
    assert x.dim() == 4 and x.shape[1] == 3 and x.is_contiguous() and x.dtype in (torch.uint8, torch.float32)
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    _tag_hbm("patches", _bytes(col) + x.numel() * x.element_size())
    check(_lib.lib().cvcs_im2col_stem(x.data_ptr(), int(x.dtype == torch.uint8), x.shape[0], x.shape[2], x.shape[3], KH, KW, stride, pad, m3, s3,
                                      col.H, col.W, col.ptr, col.ld, col.code, _stream()), "cvcs_im2col_stem")


def col2im(col: View, dx: View, KH, KW, stride, oy0, ox0, direction=1, phases=1, phase_elems=0):
    """dx [B,H,W,C] <- the transposed gather of `phases` slabs col [B,Ho,Wo,Kp] (phase_elems elements apart); oy0 / ox0: int or one per phase"""
    import ctypes as C
    oy = [oy0] * phases if isinstance(oy0, int) else list(oy0)
    ox = [ox0] * phases if isinstance(ox0, int) else list(ox0)
    assert len(oy) == len(ox) == phases
    a, b = (C.c_int * 4)(*(oy + [0] * (4 - phases))), (C.c_int * 4)(*(ox + [0] * (4 - phases)))
    _tag_hbm("patches", _bytes(col) * phases + _bytes(dx))
    check(_lib.lib().cvcs_col2im(col.ptr, col.ld, phase_elems, phases, a, b, direction, stride, KH, KW, col.B, col.H, col.W, dx.C, dx.H, dx.W,
                                 dx.ptr, dx.ld, dx.code, _stream()), "cvcs_col2im")


def phase_shuffle(ph: View, phase_elems: int, out: View, relu: bool):
    """four phase maps -> out [B,2H,2W,C] (+ReLU): slabs [4][B,H,W,C] (ph = phase 0, phase_elems = a slab) or the phases interleaved in the
    channels of one map [B,H,W,4C] (ph = that map, phase_elems = C)"""
    _tag_hbm("patches", _bytes(out) * 2)
    check(_lib.lib().cvcs_phase_shuffle(ph.ptr, ph.ld, phase_elems, ph.B, ph.H, ph.W, out.C, out.ptr, out.ld, int(relu), ph.code, _stream()),
          "cvcs_phase_shuffle")


def phase_unshuffle(g: View, relu_out: View | None, gph: View, phase_elems: int):
    _tag_hbm("patches", _bytes(g) * (3 if relu_out is not None else 2))
    check(_lib.lib().cvcs_phase_unshuffle(g.ptr, g.ld, 0 if relu_out is None else relu_out.ptr, 0 if relu_out is None else relu_out.ld, gph.B, gph.H,
                                          gph.W, g.C, gph.ptr, gph.ld, phase_elems, g.code, _stream()), "cvcs_phase_unshuffle")


def dwconv3x3(x: View, w, bias, out: View, flip=False):
    """depthwise 3x3 / pad 1 (w f32 [C,1,3,3] as stored, bias f32 [C] | None); flip: the data gradient"""
    assert w.dtype == torch.float32 and w.is_contiguous() and w.numel() == 9 * x.C
    _tag_hbm("dwconv", _bytes(x, out))
    check(_lib.lib().cvcs_dwconv3x3(x.ptr, x.ld, x.B, x.H, x.W, x.C, w.data_ptr(), _ptr(bias), int(flip), out.ptr, out.ld, x.code, _stream()),
          "cvcs_dwconv3x3")


def dwconv3x3_wgrad_rows(M: int) -> int:
    return _lib.lib().cvcs_dwconv3x3_wgrad_rows(M)


def dwconv3x3_wgrad(x: View, dy: View, part):
    _tag_hbm("dwconv", _bytes(x, dy))
    check(_lib.lib().cvcs_dwconv3x3_wgrad(x.ptr, x.ld, dy.ptr, dy.ld, x.B, x.H, x.W, x.C, part.data_ptr(), x.code, _stream()), "cvcs_dwconv3x3_wgrad")


def drop_path_scales(state: torch.Tensor, rates: torch.Tensor, B: int, out: torch.Tensor):
    """out f32 [n, B]: the per-sample factors of n DropPath calls of this step (state = [seed, step] int64 on the device)"""
    assert state.dtype == torch.int64 and rates.dtype == torch.float32 and out.dtype == torch.float32 and out.numel() == rates.numel() * B
    check(_lib.lib().cvcs_drop_path_scales(state.data_ptr(), rates.data_ptr(), rates.numel(), B, out.data_ptr(), _stream()), "cvcs_drop_path_scales")


def drop_path_scales_host(seed: int, step: int, rates, B: int):
    """cvcs_drop_path_scales replayed on the host -> f32 [n, B]"""
    import numpy as np
    n = len(rates)
    with np.errstate(over="ignore"):
        idx = np.arange(n * B, dtype=np.uint64)
        z = ((np.uint64(seed) ^ np.uint64(0x5DEECE66D)) ^ (np.uint64(step) * np.uint64(0xD6E8FEB86659FD93))) + idx * np.uint64(0x9E3779B97F4A7C15)
        z ^= z >> np.uint64(30); z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27); z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
        u = (z >> np.uint64(40)).reshape(n, B)
    out = np.ones((n, B), dtype=np.float32)
    for i, r in enumerate(rates):
        if r > 0:
            keep = np.float32(1.0) - np.float32(r)
            thr = np.uint64(int(keep * np.float32(16777216.0)))
            out[i] = np.where(u[i] < thr, np.float32(1.0) / keep, np.float32(0.0))
    return torch.from_numpy(out)


def scale_rows_add(x: View, scale, res: View | None, out: View):
    """out[b] = (res[b] if res else 0) + scale[b] * x[b]   (scale f32 [B] on the device | None = 1)"""
    _tag_hbm("residual", _bytes(x, out) + (_bytes(res) if res is not None else 0))
    check(_lib.lib().cvcs_scale_rows_add(x.ptr, x.ld, _ptr(scale), 0 if res is None else res.ptr, 0 if res is None else res.ld, x.B, x.H * x.W, x.C,
                                         out.ptr, out.ld, x.code, _stream()), "cvcs_scale_rows_add")


def sr_attention_fwd(q: View, kv: View, heads: int, out: View, lse):
    """q [B,H,W,heads*D], kv [B,Hk,Wk,2*heads*D] (keys | values), D = 32 | 64 -> out like q; lse f32 [B*heads*N]"""
    N, Nk = q.H * q.W, kv.H * kv.W
    D = q.C // heads
    assert q.C == heads * D and kv.C == 2 * q.C and kv.B == q.B and lse.numel() >= q.B * heads * N
    if _lib._recording is not None:
        _lib.pending_tag = ("sr_attention" + (f":{SCOPE}" if SCOPE else ""), 4.0 * q.B * N * Nk * q.C)
    check(_lib.lib().cvcs_sr_attention_fwd(q.ptr, q.ld, kv.ptr, kv.ld, q.B, N, Nk, heads, D, out.ptr, out.ld, lse.data_ptr(), q.code, _stream()),
          "cvcs_sr_attention_fwd")


def sr_attention_bwd_workspace(B, N, Nk, heads, head_dim=32) -> int:
    return _lib.lib().cvcs_sr_attention_bwd_workspace(B, N, Nk, heads, head_dim)


def sr_attention_bwd(q: View, kv: View, o: View, go: View, lse, heads: int, gq: View, gkv: View, workspace):
    N, Nk = q.H * q.W, kv.H * kv.W
    D = q.C // heads
    assert workspace.numel() >= sr_attention_bwd_workspace(q.B, N, Nk, heads, D)
    if _lib._recording is not None:
        _lib.pending_tag = ("sr_attention" + (f":{SCOPE}" if SCOPE else ""), 10.0 * q.B * N * Nk * q.C)
    check(_lib.lib().cvcs_sr_attention_bwd(q.ptr, q.ld, kv.ptr, kv.ld, o.ptr, o.ld, go.ptr, go.ld, lse.data_ptr(), q.B, N, Nk, heads, D, gq.ptr, gq.ld,
                                           gkv.ptr, gkv.ld, workspace.data_ptr(), q.code, _stream()), "cvcs_sr_attention_bwd")


def gather_table(items, device):
    """items: dicts with the fields of cvcs_gather_item (src / dst tensors, base, R, Cp, Rv, Cv, row = (d1, d2, s0, s1, s2), col = (...), f32_out)
    -> (device-resident table, n)"""
    arr = (_lib.GatherItem * len(items))()
    for it, d in zip(arr, items):
        it.src, it.dst, it.base = d["src"].data_ptr(), d["dst"].data_ptr(), d.get("base", 0)
        it.R, it.Cp, it.Rv, it.Cv = d["R"], d["Cp"], d["Rv"], d["Cv"]
        it.rd1, it.rd2, it.rs0, it.rs1, it.rs2 = d["row"]
        it.cd1, it.cd2, it.cs0, it.cs1, it.cs2 = d["col"]
        it.f32_out = int(d.get("f32_out", False))
        it.rv2, it.cv2 = d.get("rv2", 0), d.get("cv2", 0)
        assert d["dst"].numel() >= d["R"] * d["Cp"] and d["Rv"] <= d["R"] and d["Cv"] <= d["Cp"]
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device), len(items)


GATHER_ITEM_BYTES = 120


def gather_weights(table, n, dtype, first=0):
    check(_lib.lib().cvcs_gather_weights(table.data_ptr() + first * GATHER_ITEM_BYTES, n, dtype_code(dtype), _stream()), "cvcs_gather_weights")


def scatter_weight_grads(table, n, first=0):
    """items [first, first + n) of a table whose `src` fields point at master GRADIENT tensors and `dst` at f32 gathered-layout gradients"""
    check(_lib.lib().cvcs_scatter_weight_grads(table.data_ptr() + first * GATHER_ITEM_BYTES, n, _stream()), "cvcs_scatter_weight_grads")


def planes_from_nhwc(x: View, NC: int, planes: torch.Tensor):
    """NHWC (padded channels) -> NCHW f32 [B, NC, H, W]"""
    assert planes.dtype == torch.float32 and planes.is_contiguous() and planes.numel() == x.B * NC * x.H * x.W
    _tag_hbm("layout", _bytes(x) + planes.numel() * 4)
    check(_lib.lib().cvcs_planes_from_nhwc(x.ptr, x.ld, x.B, x.H * x.W, x.C, NC, planes.data_ptr(), x.code, _stream()), "cvcs_planes_from_nhwc")


def nhwc_from_planes(planes: torch.Tensor, NC: int, x: View):
    """NCHW f32 [B, NC, H, W] -> NHWC with zero padding channels"""
    assert planes.dtype == torch.float32 and planes.is_contiguous() and planes.numel() == x.B * NC * x.H * x.W
    _tag_hbm("layout", _bytes(x) + planes.numel() * 4)
    check(_lib.lib().cvcs_nhwc_from_planes(planes.data_ptr(), x.B, x.H * x.W, NC, x.ptr, x.ld, x.C, x.code, _stream()), "cvcs_nhwc_from_planes")


def deconv_pack(w, bias, k, p, CP, layout, wf, wd, bias4):
    """ConvTranspose2d(NC, NC, k, 2, p) master weight [NC,NC,k,k] (IOHW) -> the operands of its one-convolution form (cvcs_deconv_pack)"""
    NC = w.shape[0]
    assert w.dtype == torch.float32 and w.is_contiguous() and tuple(w.shape) == (NC, NC, k, k) and bias4.numel() == 4 * CP
    CPk = wf.shape[2] if layout == 0 else CP
    Kp = 0 if layout == 0 else wf.shape[-1]
    check(_lib.lib().cvcs_deconv_pack(w.data_ptr(), bias.data_ptr(), NC, k, p, CP, layout, CPk, Kp, wf.data_ptr(), wd.data_ptr(), bias4.data_ptr(),
                                      dtype_code(wf.dtype), _stream()), "cvcs_deconv_pack")


def deconv_unpack_grad(dw, db4, k, p, CP, layout, gw, gb):
    NC = gw.shape[0]
    Kp = 0 if layout == 0 else dw.shape[-1] if dw.dim() == 2 else dw.shape[1]
    check(_lib.lib().cvcs_deconv_unpack_grad(dw.data_ptr(), db4.data_ptr(), NC, k, p, CP, layout, Kp, gw.data_ptr(), gb.data_ptr(), _stream()),
          "cvcs_deconv_unpack_grad")


def scale_unless_one(x: torch.Tensor, scalar: torch.Tensor):
    """x *= scalar (a one-element f32 tensor ON THE DEVICE) unless it is exactly 1 - decided by the kernel, no host round trip"""
    assert x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous() and scalar.dtype == torch.float32 and scalar.numel() == 1 and scalar.device == x.device
    if x.dtype == torch.bfloat16:
        check(_lib.lib().cvcs_scale_unless_one_bf16(x.data_ptr(), x.numel(), scalar.data_ptr(), _stream()), "cvcs_scale_unless_one_bf16")
        return
    check(_lib.lib().cvcs_scale_unless_one(x.data_ptr(), x.numel(), scalar.data_ptr(), _stream()), "cvcs_scale_unless_one")



# ------------------------------------------------------------------------------------------------ MobileNetV3 pieces (csrc/mobilenet.hip)
HARDSWISH = 2          # activation code of bn_act (relu=...); its BatchNorm-backward mode is BN_MODE_HARDSWISH
BN_MODE_HARDSWISH = 3


def dwconv_rows(M: int, C_: int, dtype) -> int:
    return _lib.lib().cvcs_dwconv_rows(M, C_, dtype_code(dtype))


def dwconv(x: View, w: torch.Tensor, K, stride, pad, dil, y: View, stats=None):
    """depthwise K x K convolution, w f32 [K*K][C] tap-major; stats = (sum, m2, cnt) partial BatchNorm statistics of dwconv_rows(M, C, dtype) rows"""
    assert w.dtype == torch.float32 and w.is_contiguous() and w.numel() == K * K * x.C and (y.B, y.C) == (x.B, x.C)
    assert (y.H, y.W) == conv_out_hw(x.H, x.W, K, K, stride, pad, dil)
    if stats is not None:
        rows = dwconv_rows(y.B * y.H * y.W, y.C, y.t.dtype)
        assert stats[0].numel() >= rows * y.C and stats[1].numel() >= rows * y.C and stats[2].numel() >= rows
    _tag_hbm("dwconv", _bytes(x, y))
    s = [0, 0, 0] if stats is None else [t.data_ptr() for t in stats]
    check(_lib.lib().cvcs_dwconv(x.ptr, x.ld, x.B, x.H, x.W, x.C, w.data_ptr(), K, stride, pad, dil, y.ptr, y.ld, y.H, y.W, *s, x.code, _stream()),
          "cvcs_dwconv")


def dwconv_dgrad(dy: View, w: torch.Tensor, K, stride, pad, dil, dx: View):
    assert w.dtype == torch.float32 and w.is_contiguous() and w.numel() == K * K * dy.C and (dx.B, dx.C) == (dy.B, dy.C)
    _tag_hbm("dwconv", _bytes(dy, dx))
    check(_lib.lib().cvcs_dwconv_dgrad(dy.ptr, dy.ld, dy.B, dy.H, dy.W, dy.C, w.data_ptr(), K, stride, pad, dil, dx.ptr, dx.ld, dx.H, dx.W, dy.code,
                                       _stream()), "cvcs_dwconv_dgrad")


def dwconv_wgrad_rows(M: int, C_: int, K: int, dtype) -> int:
    return _lib.lib().cvcs_dwconv_wgrad_rows(M, C_, K, dtype_code(dtype))


def dwconv_wgrad(x: View, dy: View, K, stride, pad, dil, part: torch.Tensor):
    """partial rows part[dwconv_wgrad_rows(M, C)][K*K][C] of the depthwise weight gradient (colsum_finalize sums them)"""
    rows = dwconv_wgrad_rows(dy.B * dy.H * dy.W, dy.C, K, dy.t.dtype)
    assert part.dtype == torch.float32 and part.numel() >= rows * K * K * dy.C and (x.B, x.C) == (dy.B, dy.C)
    _tag_hbm("dwconv", _bytes(x, dy))
    check(_lib.lib().cvcs_dwconv_wgrad(x.ptr, x.ld, dy.ptr, dy.ld, x.B, x.H, x.W, x.C, K, stride, pad, dil, dy.H, dy.W, part.data_ptr(), x.code, _stream()),
          "cvcs_dwconv_wgrad")
    return rows


def se_scale(x: View, s: View, out: View, add: View | None = None, add_scale: float = 0.0):
    """out[b,p,c] = x[b,p,c] * s[b,c] (+ add[b,c] * add_scale)"""
    assert (s.B, s.H, s.W, s.C) == (x.B, 1, 1, x.C) and (out.B, out.H, out.W, out.C) == (x.B, x.H, x.W, x.C)
    _tag_hbm("se", _bytes(x, out))
    check(_lib.lib().cvcs_se_scale(x.ptr, x.ld, s.ptr, s.ld, 0 if add is None else add.ptr, 0 if add is None else add.ld, add_scale, x.B, x.H * x.W, x.C,
                                   out.ptr, out.ld, x.code, _stream()), "cvcs_se_scale")


def image_dot(a: View, x: View, out: View):
    """out [B,1,1,C] = per-image sum over the pixels of a * x"""
    assert (out.B, out.H, out.W, out.C) == (x.B, 1, 1, x.C) and (a.B, a.H, a.W, a.C) == (x.B, x.H, x.W, x.C)
    _tag_hbm("se", _bytes(a, x))
    check(_lib.lib().cvcs_image_dot(a.ptr, a.ld, x.ptr, x.ld, x.B, x.H * x.W, x.C, out.ptr, out.ld, x.code, _stream()), "cvcs_image_dot")


def hardsigmoid(t: View, out: View, g: View | None = None):
    """out = hardsigmoid(t), or with g its backward g * hardsigmoid'(t)"""
    check(_lib.lib().cvcs_hardsigmoid(t.ptr, t.ld, 0 if g is None else g.ptr, 0 if g is None else g.ld, t.B * t.H * t.W, t.C, out.ptr, out.ld, t.code,
                                      _stream()), "cvcs_hardsigmoid")


def bn_add(y: View, scale, shift, res: View | None, out: View):
    """out = scale*y + shift (+ res): BatchNorm apply of a linear bottleneck projection with the residual add, no activation"""
    _tag_hbm("residual", _bytes(y, out) + (0 if res is None else _bytes(res)))
    check(_lib.lib().cvcs_bn_add(y.ptr, y.ld, scale.data_ptr(), shift.data_ptr(), 0 if res is None else res.ptr, 0 if res is None else res.ld,
                                 y.B * y.H * y.W, y.C, out.ptr, out.ld, y.code, _stream()), "cvcs_bn_add")
