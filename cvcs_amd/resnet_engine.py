"""Launch plan of the ResNet-encoder U-Nets (BASELINE.json configs 1, 2, 5: "ResNet18-UNet", "ResNet50-UNet") on the HIP
kernels.

The reference has no such model (its factory S/utils.py:174-195 is the seam they plug into, see nets.ResnetUnet); the
architecture (the CPU checker states it in plain torch) is: ResNet v1.5 encoder (7x7/s2 stem, 3x3/s2
max-pool, BasicBlock / Bottleneck stages) and a bilinear-upsample decoder of (conv3x3 -> BN -> ReLU) pairs with the encoder
features concatenated, 1x1 head.

MI355X-first choices:
  * launch plan, not a module tree: the first pass for an input shape runs eagerly under `_lib.Recording`, every later
    step REPLAYS the recorded list of C-ABI launches (about a thousand per training step for ResNet-50) with no Python in
    between - the host side of a step is one loop over pre-built argument tuples;
  * activations NHWC bf16 (f32 on the parity path); `torch.cat` never runs: encoder features and the up-sampled decoder
    tensor are written by their producers into channel ranges of one buffer;
  * BatchNorm batch statistics come out of the conv epilogues; the block tail (BN apply + shortcut (+ its BN) + add +
    ReLU) is one pass; its backward (sum of up to three incoming gradients x ReLU mask) is one pass that also scatters
    the half-resolution gradient of a stride-2 projection shortcut;
  * the 7x7/s2 stem runs as a 7x1 filter over "virtual pixels" (64-byte windows of the 4-channel padded tile): seven
    K-steps instead of 49 and no im2col buffer (cvcs_conv_desc.aniso);
  * strided 3x3 data gradients = zero-dilated gradient through the stride-1 halo kernel; strided 1x1 data gradients are
    computed at half resolution and scattered by the consumer;
  * eval mode folds every BatchNorm into its conv's epilogue.
"""
from __future__ import annotations

import collections
import contextlib
import os

import torch

from . import _lib, ops
from .engine import _BN
from .ops import View

ARCHS = {
    "resnet18": ("basic", (2, 2, 2, 2), (64, 128, 256, 512)),
    "resnet34": ("basic", (3, 4, 6, 3), (64, 128, 256, 512)),
    "resnet50": ("bottleneck", (3, 4, 6, 3), (256, 512, 1024, 2048)),
    "resnet101": ("bottleneck", (3, 4, 23, 3), (256, 512, 1024, 2048)),
}


def stage_plan(arch, output_stride=32):
    """[(stage, block, stride, dilation)]: torchvision's `replace_stride_with_dilation` rule (the DeepLab encoders) - once the
    output stride is reached a stage's stride-2 block keeps stride 1 (and the dilation so far), its later blocks double it"""
    _, depths, _ = ARCHS[arch]
    plan, cur, dil = [], 4, 1
    for s, n in enumerate(depths, start=1):
        first_stride, first_dil = 1, dil
        if s > 1:
            if cur >= output_stride:
                dil *= 2
            else:
                first_stride, cur = 2, cur * 2
        for b in range(n):
            plan.append((s, b, first_stride if b == 0 else 1, first_dil if b == 0 else dil))
    return plan
# decoder widths: the common public U-Net-on-ResNet layout (segmentation_models_pytorch's default (256, 128, 64, 32, 16)).  Rounds 1-2 kept the
# last two stages at 64 channels because no kernel tiled output channels below 64; the 3x3 halo kernel now masks 16 / 32 / 48-channel tiles,
# the head takes 16 / 32 / 64 input channels (config key `decoder_channels` restores any other widths, e.g. [256, 128, 64, 64, 64])
DECODER_CHANNELS = (256, 128, 64, 32, 16)
GN_GROUPS = 32


class Act:
    """an activation and the gradients its consumers produce for it during backward: [(view, half_resolution)]"""

    def __init__(self, v: View):
        self.v = v
        self.grads = []


class GridView:
    """an activation stored in sub-grid layout L_d, presented in the ordinary pixel order (tests read relu_order through .torch())"""

    def __init__(self, v: View, d: int):
        self.v, self.d = v, d

    def torch(self):
        t, d = self.v.torch(), self.d
        Bd, Hd, Wd, C_ = t.shape
        return t.reshape(Bd // (d * d), d, d, Hd, Wd, C_).permute(0, 3, 1, 4, 2, 5).reshape(Bd // (d * d), Hd * d, Wd * d, C_)


class Unit:
    """one conv (+ BatchNorm): input view, conv output y, geometry"""

    def __init__(self, x, y, conv, bn, k, stride, pad, virt=False, dil=1):
        self.x, self.y, self.conv, self.bn, self.k, self.stride, self.pad, self.virt = x, y, conv, bn, k, stride, pad, virt
        self.dil = dil


class ResNetUNetEngine:
    def __init__(self, arch: str, num_classes: int, dtype: torch.dtype, device, decoder_channels=DECODER_CHANNELS,
                 decoder_norm="bn_relu", fp8=False):
        assert arch in ARCHS and decoder_norm in ("bn_relu", "gn_silu")
        assert not fp8 or dtype == torch.bfloat16, "the fp8 convolutions live beside bf16 storage (precision 'fp8' = mixed bf16 / fp8)"
        # fp8 = BASELINE configs[4] "mixed bf16/fp8 convs": the 3x3 / stride 1 convolutions whose channel counts are multiples of 128 run
        # forward AND data gradient on the block-scaled fp8 MFMA (csrc/conv_fp8.hip); everything else, the weight gradients included, bf16
        self.fp8 = bool(fp8)
        self.f8 = None                   # ops.Fp8Slots: per-tensor scales (delayed scaling)
        self._q8 = {}                    # bf16 buffer address -> (fp8 buffer, slot, format) of the tensors an fp8 convolution reads
        self._train_pass = True
        # evaluation passes run the bf16 kernels by default: in eval mode the BatchNorms are folded into the conv epilogues, so no producer
        # pass exists to write the fp8 images for free, and fp8 is a TRAINING-throughput feature (CVCS_FP8_EVAL=1: fp8 with the trained scales)
        self.fp8_eval = os.environ.get("CVCS_FP8_EVAL", "0") == "1"
        self.decoder_norm = decoder_norm
        self.arch, self.NC, self.dtype, self.dev = arch, num_classes, dtype, torch.device(device)
        self.kind, self.depths, self.widths = ARCHS[arch]
        self.dec = tuple(decoder_channels)
        assert len(self.dec) == 5 and all(c % 16 == 0 for c in self.dec) and self.dec[-1] in (16, 32, 64), \
            "decoder widths: five multiples of 16, the last one 16 / 32 / 64 (the 1x1 head kernels)"
        self.head_name = "segmentation_head.0"
        self.code = ops.dtype_code(dtype)
        self.P = self.G = self.Bf = None
        self.shape = None
        self.on_backward_begin = None    # data-parallel hooks (cvcs_amd.parallel)
        self.on_grad_ready = None
        self.sync_bn = None
        self._bufs = {}
        self._rec = {}                   # "fwd_train" | "fwd_eval" | "bwd" -> _lib.Recording
        self._tape = None
        self._saved_train = False
        self._last_act = {}
        self._shape_cache = collections.OrderedDict()
        self.fuse_bn_bwd = os.environ.get("CVCS_FUSE_BN_BWD", "1") == "1"   # see _fusable
        self.gather_shortcut = os.environ.get("CVCS_GATHER_SHORTCUT", "1") == "1"   # stride-2 projection shortcuts on compact even-pixel maps (_block)
        # bf16: the 1x1 / stride-2 projection shortcuts read the even pixels of their input IN PLACE (taps kernel forward, GEMM weight gradient with
        # a strided x) - no gathered copy of the input (the f32 parity path and widths off the 128-channel tile keep the round-3 gather)
        self.strided_shortcut = os.environ.get("CVCS_STRIDED_SHORTCUT", "1") == "1"
        self.fuse_tail_bn = os.environ.get("CVCS_FUSE_TAIL_BN", "1") == "1"  # the residual tails' BatchNorm reduce passes on relu_bwd_sum (_block)
        # bottleneck tails without a stored conv3 output: BatchNorm statistics from the Gram matrix of conv3's input, BatchNorm + shortcut +
        # ReLU in conv3's epilogue, the BatchNorm backward folded into the weight- / data-gradient GEMMs (_block, csrc/bn_gram.hip)
        self.gram_bn = os.environ.get("CVCS_GRAM_BN", "1") == "1"
        self.mask_bits = os.environ.get("CVCS_MASK_BITS", "1") == "1"       # ReLU masks of the Gram tails as bits (_block, _dgrad_into_tail)
        self.fuse_upsample = os.environ.get("CVCS_FUSE_UPSAMPLE", "0") == "1"   # opt-in (validated at kernel level only): see the decoder loop of _plan
        self.thin_bn_bwd = os.environ.get("CVCS_THIN_BN_BWD", "1") == "1"     # the thin kernel's data-gradient launches carry the next BatchNorm's reduce (_fusable)
        self.fuse_stem_bwd = os.environ.get("CVCS_FUSE_STEM_BWD", "1") == "1"   # see _stem
        self.gram_ds = os.environ.get("CVCS_GRAM_DS", "1") == "1"     # a same-resolution projection shortcut joins its Gram tail as a second GEMM source (_block)
        self.fuse_tail_dz = os.environ.get("CVCS_FUSE_TAIL_DZ", "1") == "1"   # a Gram tail's dz out of the consumer block's first data-gradient launch (_tail_fusable)
        self.gram_512_pixels = int(os.environ.get("CVCS_GRAM_512_PIXELS", "65536"))   # 512-channel conv3 inputs take it from this many pixels on
        self.gram_max_m = int(os.environ.get("CVCS_GRAM_MAX_M", "256"))   # widest conv3 input that takes the path (the finalizes are O(C m^2))
        # Weight gradients are off the critical chain of backward (dy -> data gradient -> BatchNorm backward of the layer below): they
        # are recorded on the plan's side lane and replayed on a second HIP stream, so the MFMA-bound weight-gradient kernels run
        # beside the HBM-bound BatchNorm / residual passes (CVCS_OVERLAP_WGRAD=0: one stream, in program order)
        self.overlap_wgrad = os.environ.get("CVCS_OVERLAP_WGRAD", "0") == "1"
        self.grid_dilated = os.environ.get("CVCS_GRID_DILATED", "1") == "1"   # dilated stages in sub-grid layout (see _stages)
        self._grid = 1
        self._side_stream = None
        self._dy_reader = {}             # dy scratch buffer -> index of the side launch that read it last (this recording)
        self._dy_toggle = 0
        self.keep_all = False            # tests: every backward intermediate in its own buffer (no scratch reuse) + a registry
        self.fwd_serial = 0              # forward passes so far (a nets.LazyLogits handle is current while it matches)
        self._head_pending = self._head_done = False
        self._head_rows = 0
        self.bwd_units = {}
        self.bwd_gram = {}               # keep_all: the Gram-path tails (dz, data gradient) by conv name

    # ------------------------------------------------------------------------------------------------ binding
    def bind(self, params, grads, buffers):
        self.P, self.G, self.Bf = params, grads, buffers
        base = min(g.data_ptr() for g in grads.values())
        self._goff = {k: (g.data_ptr() - base) // 4 for k, g in grads.items()}
        dev, dt = self.dev, self.dtype
        self.packed = {}
        entries = []
        for name, w in params.items():
            if w.dim() != 4 or name in ("encoder.conv1.weight", self.head_name + ".weight"):
                continue
            cout, cin, kh, kw = w.shape
            kg = 64 // (2 if dt == torch.bfloat16 else 4)      # channels per 64-byte K-step: a thinner contraction axis is zero-padded to one
            pk = dict(wf=torch.zeros(kh * kw, cout, max(cin, kg), dtype=dt, device=dev), wd=torch.zeros(kh * kw, cin, max(cout, kg), dtype=dt, device=dev))
            self.packed[name[:-len(".weight")]] = pk
            entries.append((w, pk["wf"], pk["wd"]))
        self._pack_table = ops.pack_table(entries, dev)
        if self.fp8:
            self.f8 = ops.Fp8Slots(dev)
            for name, pk in self.packed.items():
                _, cout, cin = pk["wf"].shape
                if pk["wf"].shape[0] == 9 and cout % 128 == 0 and cin % 128 == 0:      # (which of them run fp8 is decided per launch: _fp8_ok)
                    pk["wf8"], pk["wd8"] = (torch.empty(t.shape, dtype=torch.uint8, device=dev) for t in (pk["wf"], pk["wd"]))
        self.stem_wf = torch.empty(7, 64, 32, dtype=dt, device=dev)
        self.stem_dw_tmp = torch.empty(64 * 32 * 7, dtype=torch.float32, device=dev)
        self.bn = {n[:-len(".weight")]: _BN(p.numel(), dev) for n, p in params.items() if p.dim() == 1 and n.endswith(".weight")}
        cmax = max(max(self.widths), 64)
        self.one = torch.ones(cmax, dtype=torch.float32, device=dev)
        self.zero = torch.zeros(cmax, dtype=torch.float32, device=dev)
        self._rec, self._bufs, self.shape = {}, {}, None
        self._shape_cache = collections.OrderedDict()
        self.gn = {}                     # GroupNorm state per layer (allocated per batch size at plan time)

    def enable_sync_bn(self, sync):
        """exact data-parallel mode for every network on this engine (ResNet-UNets, DeepLab, Swin + UPerNet):
        `sync` has .world and .all_reduce(tensor) (sum over ranks, in place, ordered on the current stream).  Each BatchNorm's batch moments
        (forward) and its two backward sums are summed over ranks, so N ranks x B tiles normalise like one process on N*B tiles.  A
        BatchNorm's statistics depend on the previous layer's normalised output, so these are one small collective per BatchNorm and
        direction (<= 3 x 2048 doubles), issued from host callbacks recorded INSIDE the launch plans at the point of use."""
        self.sync_bn = sync
        cmax = max(p.numel() for n, p in self.P.items() if p.dim() == 1)
        self._sync_mom = torch.empty(3 * cmax, dtype=torch.float64, device=self.dev)
        self._sync_sums = torch.empty(2 * cmax, dtype=torch.float32, device=self.dev)
        self._rec = {}            # plans recorded without the exchanges are stale
        self._shape_cache.clear()

    def refresh_weights(self, train=True):
        ops.pack_conv_weights(self._pack_table[0], self._pack_table[1], self.dtype)
        ops.pack_stem_weight(self.P["encoder.conv1.weight"], self.stem_wf)
        if self.fp8:
            # e4m3 images of the packed bf16 weights, one scale per layer (forward and data-gradient image hold the same values)
            for name, pk in self.packed.items():
                if "wf8" in pk:
                    i = self.f8.slot("w:" + name, _lib.E4M3)
                    for src, dst in ((pk["wf"], pk["wf8"]), (pk["wd"], pk["wd8"])):
                        rows, k = src.shape[0] * src.shape[1], src.shape[2]
                        ops.quantize_fp8(View(src.view(1, rows, 1, k), 0, k), View(dst.view(1, rows, 1, k), 0, k), _lib.E4M3, self.f8, i,
                                         take_amax=train and src is pk["wf"])

    def _fp8_ok(self, conv, x: View, k, stride, dil, virt=False):
        """does this convolution run on the fp8 kernel?  3x3 / stride 1 / no dilation on maps of at least 8 pixels, both channel counts
        multiples of 128 (one MFMA K-step = 128 input channels; 128 output channels per workgroup)"""
        pk = self.packed.get(conv)
        return (self.fp8 and not virt and pk is not None and "wf8" in pk and k == 3 and stride == 1 and dil == 1 and x.H >= 8 and x.W >= 8 and
                x.C % 128 == 0 and (self._train_pass or self.fp8_eval))

    # The fp8 image of a tensor is written by the pass that PRODUCES the tensor (bn_act / the block tail / the bilinear up-sampling /
    # the BatchNorm-backward apply: ops.*(..., q8=...)), not by a sweep of its own.  A tensor that an fp8 convolution will read is registered
    # before its producers run; producers and the consumer look it up by the bf16 buffer's address.
    def _register_q8(self, t: torch.Tensor, tag, fmt=_lib.E4M3, buf=None):
        q = self._buf("f8." + (buf or tag), tuple(t.shape), torch.uint8)
        self._q8[t.data_ptr()] = (q, self.f8.slot(tag, fmt), fmt)

    def _q8_of(self, v: View, train=True):
        """the q8 argument of a producer that writes the view v, or None"""
        if not self.fp8 or v is None:
            return None
        r = self._q8.get(v.t.data_ptr())
        if r is None:
            return None
        q, i, fmt = r
        return (View(q, v.off, v.C), fmt, self.f8, i, train)

    def _fp8_input(self, x: View, tag, fmt, train, buf=None):
        """(fp8 view, slot) of the convolution input x.  Registered tensors were written by their producers; a slot WITHOUT history is
        calibrated here, outside the recording: the producers have already fed its amax (which does not depend on the scale), so the scale
        is set from it and the image is taken again once with that scale - later steps run on the delayed scale alone.  Unregistered
        tensors (no fused producer) get a quantisation pass of their own."""
        r = self._q8.get(x.t.data_ptr())
        if r is None:
            q = self._buf("f8." + (buf or tag), tuple(x.t.shape), torch.uint8)
            i = self.f8.slot(tag, fmt)
            qv = View(q, x.off, x.C)
            ops.quantize_fp8(x, qv, fmt, self.f8, i, take_amax=train)
            return qv, i
        q, i, _ = r
        qv = View(q, x.off, x.C)
        if i in self.f8.fresh:
            h, st = _lib._load(), torch.cuda.current_stream().cuda_stream
            _lib.check(h.cvcs_fp8_update_scales(self.f8.ptr(i), 1, ops.FP8_MARGIN, st), "cvcs_fp8_update_scales")
            _lib.check(h.cvcs_quantize_fp8(x.ptr, x.ld, x.B * x.H * x.W, x.C, qv.ptr, qv.ld, fmt, self.f8.ptr(i), 1, st), "cvcs_quantize_fp8")
            self.f8.fresh.discard(i)
        return qv, i

    # ------------------------------------------------------------------------------------------------ buffers
    def _buf(self, name, shape, dtype=None):
        key = (name, tuple(shape), dtype or self.dtype)
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = torch.empty(shape, dtype=dtype or self.dtype, device=self.dev)
        return t

    def _zbuf(self, name, shape, dtype=None):
        """like _buf, zero-filled at allocation (buffers whose padding must read as zero)"""
        key = (name, tuple(shape), dtype or self.dtype)
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = torch.zeros(shape, dtype=dtype or self.dtype, device=self.dev)
        return t

    def _act(self, name, B, H, W, C_):
        return self._buf(name, (B, H, W, C_))

    def _scratch(self, tag, n, dtype=torch.float32):
        return self._buf(f"scratch.{tag}", (int(n),), dtype)

    # ------------------------------------------------------------------------------------------------ forward pieces
    def _unit(self, x: View, conv, bn, k, stride, pad, train, act_out: View | None, relu=True, virt=False, dil=1) -> Unit:
        """conv (+ batch statistics) -> BatchNorm finalize [-> apply (+ReLU) into act_out].  act_out None: the caller applies
        the BatchNorm itself (block tail).  Eval mode: the BatchNorm is folded into the conv epilogue; y then holds bn(conv)."""
        B = x.B
        w = self.P[conv + ".weight"]
        cout = w.shape[0]
        ops.SCOPE = "enc" if conv.startswith("encoder.") else "dec"
        if virt:
            Ho, Wo = x.H // 2, (x.W - 8) // 2
            wf = self.stem_wf
            kh, kw = 7, 1
        else:
            Ho, Wo = ops.conv_out_hw(x.H, x.W, k, k, stride, pad, dil)
            wf = self.packed[conv]["wf"]
            kh = kw = k
        st = self.bn[bn]
        M = B * Ho * Wo
        if (bn + ".running_mean") not in self.Bf:
            return self._unit_gn(x, conv, bn, wf, k, stride, pad, dil, Ho, Wo, cout, act_out)
        if not train:
            ops.bn_finalize(None, 0, M, cout, self.P[bn + ".weight"], self.P[bn + ".bias"], self.Bf[bn + ".running_mean"],
                            self.Bf[bn + ".running_var"], False, st.scale, st.shift, None, None)
            out = act_out if act_out is not None else ops.view(self._act(conv + ".y", B, Ho, Wo, cout))
            if self._fp8_ok(conv, x, k, stride, dil, virt):
                x8, ix = self._fp8_input(x, "a:" + conv, _lib.E4M3, False)
                ops.conv3x3_fp8(x8, _lib.E4M3, self.packed[conv]["wf8"], out, self.f8, ix, self.f8.slot("w:" + conv, _lib.E4M3),
                                relu=relu and act_out is not None, pre_affine=(st.scale, st.shift))
            else:
                ops.conv2d(x, wf, None, out, kh, kw, stride, pad, dil, relu=relu and act_out is not None, pre_affine=(st.scale, st.shift),
                           virt=virt or None)
            return Unit(x, out, conv, bn, k, stride, pad, virt, dil)
        y = ops.view(self._act(conv + ".y", B, Ho, Wo, cout))
        f8 = self._fp8_ok(conv, x, k, stride, dil, virt)
        rows = ops.fp8_stat_rows(x) if f8 else ops.conv_stat_rows(x, cout, kh, kw, stride, pad, dil, virt=virt or None)
        stats = (self._scratch("stat_sum", rows * cout), self._scratch("stat_m2", rows * cout), self._scratch("stat_cnt", rows))
        if f8:
            x8, ix = self._fp8_input(x, "a:" + conv, _lib.E4M3, True)
            ops.conv3x3_fp8(x8, _lib.E4M3, self.packed[conv]["wf8"], y, self.f8, ix, self.f8.slot("w:" + conv, _lib.E4M3), stats=stats)
        else:
            ops.conv2d(x, wf, None, y, kh, kw, stride, pad, dil, stats=stats, virt=virt or None)
        need = ops.bn_finalize_workspace_floats(rows, cout)
        if self.sync_bn is None:
            ops.bn_finalize(stats, rows, M, cout, self.P[bn + ".weight"], self.P[bn + ".bias"], self.Bf[bn + ".running_mean"],
                            self.Bf[bn + ".running_var"], True, st.scale, st.shift, st.mean, st.invstd,
                            workspace=self._scratch("bn_ws", max(need, 4)))
        else:
            # batch statistics over the tiles of ALL ranks: this rank's (n, sum, sum of squares) per channel in f64, one sum-all-reduce
            mom = self._sync_mom[:3 * cout]
            ops.bn_moments(stats, rows, cout, mom, workspace=self._scratch("bn_ws", max(need, 4)))
            self._host(lambda mom=mom: self.sync_bn.all_reduce(mom))
            ops.bn_finalize_moments(mom, cout, self.P[bn + ".weight"], self.P[bn + ".bias"], self.Bf[bn + ".running_mean"],
                                    self.Bf[bn + ".running_var"], st.scale, st.shift, st.mean, st.invstd)
        if act_out is not None:
            ops.bn_act(y, st.scale, st.shift, relu, act_out, q8=self._q8_of(act_out))
        u = Unit(x, y, conv, bn, k, stride, pad, virt, dil)
        u.fp8 = f8
        self.units[conv] = (u, act_out)     # persistent views of this shape's plan (layer-wise parity tests read them)
        if act_out is not None and relu:
            self.relu_order.append(act_out if self._grid == 1 else GridView(act_out, self._grid))
        return u

    def _unit_gn(self, x, conv, gn, wf, k, stride, pad, dil, Ho, Wo, cout, act_out: View):
        """conv -> GroupNorm(32) -> SiLU (train and eval alike: a GroupNorm has no running statistics): one streaming pass for
        the per-(image, channel) sums, a tiny finalize, ONE pass that normalises, scales and applies the SiLU"""
        B, HW = x.B, Ho * Wo
        y = ops.view(self._act(conv + ".y", B, Ho, Wo, cout))
        ops.conv2d(x, wf, None, y, k, k, stride, pad, dil)
        st = self.gn.setdefault(gn, {n: torch.empty(B * cout, dtype=torch.float32, device=self.dev) for n in ("scale", "shift", "ca", "cb", "cc")})
        if "mean" not in st:
            st["mean"], st["invstd"] = (torch.empty(B * GN_GROUPS, dtype=torch.float32, device=self.dev) for _ in range(2))
            st["tmp"] = torch.empty(B * 2 * cout, dtype=torch.float32, device=self.dev)
        part = self._scratch("gn_part", B * ops.gn_rows(HW) * 2 * cout)
        ops.gn_stats(y, part)
        ops.gn_finalize(part, B, HW, cout, GN_GROUPS, self.P[gn + ".weight"], self.P[gn + ".bias"], st["scale"], st["shift"], st["mean"],
                        st["invstd"])
        ops.gn_act_fwd(y, st["scale"], st["shift"], True, act_out)
        u = Unit(x, y, conv, gn, k, stride, pad, False, dil)
        u.gn = True
        self.units[conv] = (u, act_out)
        return u

    def _tail(self, u3: Unit, ud: Unit | None, h: Act, out: View, train):
        s3 = self.bn[u3.bn]
        C_ = out.C
        if train:
            q8 = self._q8_of(out)
            if ud is not None:
                sd = self.bn[ud.bn]
                ops.bn_add_act(u3.y, s3.scale, s3.shift, ud.y, sd.scale, sd.shift, out, q8=q8)
            else:
                ops.bn_add_act(u3.y, s3.scale, s3.shift, h.v, None, None, out, q8=q8)
        else:   # the BatchNorms are already inside the conv epilogues
            ops.bn_add_act(u3.y, self.one[:C_], self.zero[:C_], ud.y if ud is not None else h.v, None, None, out)

    def _block(self, h: Act, p, stride, out: View, train, tape, dil=1) -> Act:
        B = h.v.B
        has_ds = (p + ".downsample.0.weight") in self.P
        if self.kind == "basic":
            w = self.P[p + ".conv1.weight"].shape[0]
            Ho, Wo = h.v.H // stride, h.v.W // stride
            a1 = Act(ops.view(self._act(p + ".a1", B, Ho, Wo, w)))
            u1 = self._unit(h.v, p + ".conv1", p + ".bn1", 3, stride, dil, train, a1.v, dil=dil)
            ut = self._unit(a1.v, p + ".conv2", p + ".bn2", 3, 1, dil, train, None, dil=dil)
            chain = [(u1, h, a1)]
            last_in = a1
        else:
            mid = self.P[p + ".conv1.weight"].shape[0]
            Ho, Wo = h.v.H // stride, h.v.W // stride
            a1 = Act(ops.view(self._act(p + ".a1", B, h.v.H, h.v.W, mid)))
            a2 = Act(ops.view(self._act(p + ".a2", B, Ho, Wo, mid)))
            if train and self._fp8_ok(p + ".conv2", a1.v, 3, stride, dil):
                self._register_q8(a1.v.t, "a:" + p + ".conv2")
            u1 = self._unit(h.v, p + ".conv1", p + ".bn1", 1, 1, 0, train, a1.v)
            u2 = self._unit(a1.v, p + ".conv2", p + ".bn2", 3, stride, dil, train, a2.v, dil=dil)
            gram = train and self._gram_ok(p, a2.v, out)
            ut = self._unit_gram(a2.v, p + ".conv3", p + ".bn3") if gram else self._unit(a2.v, p + ".conv3", p + ".bn3", 1, 1, 0, train, None)
            chain = [(u1, h, a1), (u2, a1, a2)]
            last_in = a2
        ds_w = self.P[p + ".downsample.0.weight"].shape[0] if has_ds else 0
        in_place = self.strided_shortcut and self.dtype == torch.bfloat16 and ds_w % 128 == 0 and h.v.H % 2 == 0 and h.v.W % 2 == 0
        if has_ds and stride == 2 and self.gather_shortcut and not in_place:
            # the projection shortcut of a down-sampling block reads the even pixels only: gather them once (a quarter of h) and the 1x1 /
            # stride-2 conv becomes a plain GEMM - forward on the taps kernel, weight gradient on the GEMM kernel instead of the generic ones
            hs = ops.view(self._act(p + ".hs", B, Ho, Wo, h.v.C))
            ops.SCOPE = "enc"
            ops.im2col(h.v, hs, 1, 1, 2, 0, 0)
            ud = self._unit(hs, p + ".downsample.0", p + ".downsample.1", 1, 1, 0, train, None)
            ud.half_out = True            # its data gradient lives at half the resolution of h (scattered by the consumer)
        elif has_ds and stride == 1 and getattr(ut, "gram", False) and self.gram_ds and self._gram_ok(p, h.v, out, bn=".downsample.1"):
            # same-resolution projection (layer1.0): its input is as narrow as conv3's - statistics from the Gram matrix of h, the conv itself the
            # second source of the tail's GEMM: no projection output, no pass of its BatchNorm in either direction
            ud = self._unit_gram(h.v, p + ".downsample.0", p + ".downsample.1")
        else:
            ud = self._unit(h.v, p + ".downsample.0", p + ".downsample.1", 1, stride, 0, train, None) if has_ds else None
        # the ReLU mask of a Gram tail's output as bits (one byte per 8 channels): what the consumer block's tail-fused data gradient reads
        # instead of the output itself (_dgrad_into_tail)
        mbits = None
        if getattr(ut, "gram", False) and self.mask_bits and train and out.C % 128 == 0 and self._grid == 1:
            mbits = self._buf(p + ".mbits", (out.B * out.H * out.W * out.C // 8,), torch.uint8)
        if getattr(ut, "gram", False) and ud is not None and getattr(ud, "gram", False):
            s3, sd = self.bn[ut.bn], self.bn[ud.bn]
            w3, wdn = self.packed[ut.conv]["wf"], self.packed[ud.conv]["wf"]
            w2s = self._buf(ut.conv + ".w2s", (1, out.C, w3.shape[2] + wdn.shape[2]))
            b2s = self._buf(ut.conv + ".b2s", (out.C,), torch.float32)
            ops.bn_gram_fold(w3, s3.scale, s3.shift, wdn, sd.scale, sd.shift, w2s, b2s)
            ops.SCOPE = "enc"
            ut.out, ut.res, ut.res_bn, ut.res_gram = out, None, None, ud      # (tests)
            ops.conv2d(a2.v, w2s, b2s, out, 1, 1, relu=True, x2=h.v, mask_bits_out=mbits)
        elif getattr(ut, "gram", False):
            # conv3 -> BatchNorm -> + shortcut (its own BatchNorm for a projection) -> ReLU in ONE launch: the conv output is never stored
            s3 = self.bn[ut.bn]
            ops.SCOPE = "enc"
            ut.out, ut.res, ut.res_bn = out, (ud.y if ud is not None else h.v), (ud.bn if ud is not None else None)   # (tests)
            if ud is not None:
                sd = self.bn[ud.bn]
                ops.conv2d(a2.v, self.packed[ut.conv]["wf"], None, out, 1, 1, relu=True, pre_affine=(s3.scale, s3.shift), res=ud.y, res_affine=(sd.scale, sd.shift),
                           mask_bits_out=mbits)
            else:
                ops.conv2d(a2.v, self.packed[ut.conv]["wf"], None, out, 1, 1, relu=True, pre_affine=(s3.scale, s3.shift), res=h.v, mask_bits_out=mbits)
        else:
            self._tail(ut, ud, h, out, train)
        self.relu_order.append(out if self._grid == 1 else GridView(out, self._grid))
        o = Act(out)
        if getattr(ut, "gram", False):
            o.tail_dz, o.dz_ready = p + ".dz", None      # (its tail backward needs no BatchNorm reduce: the consumer's data gradient may write dz, _tail_fusable)
            o.mask_bits = mbits
        if train:
            def bwd():
                dz = ops.view(self._act(p + ".dz", out.B, out.H, out.W, out.C))
                red_t = red_d = None
                if getattr(ut, "gram", False):
                    # no pass over a conv3 output: the tail's backward is the gradient sum under the ReLU mask (+ the reduce pass of the
                    # projection shortcut's BatchNorm, which still has a stored output)
                    if o.dz_ready is not None:
                        assert not o.grads
                        dz = o.dz_ready           # written by the consumer block's first data-gradient launch
                    elif ud is not None and self.fuse_tail_bn and not getattr(ud, "gram", False):
                        n_ = ops.bn_bwd_rows(out.B * out.H * out.W) * out.C
                        pz = [self._scratch(f"tz{i}", n_) for i in range(2)]
                        rows_t = ops.relu_bwd_sum_bn(o.v, o.grads, dz, [(ud.y, self.bn[ud.bn].mean, self.bn[ud.bn].invstd)], pz[0], pz[1:2])
                        red_d = (pz[0], pz[1], rows_t)
                    else:
                        ops.relu_bwd_sum(o.v, o.grads, dz)
                    last_in.grads.append((self._gram_bwd(ut, dz, p + ".g_t"), False))
                    self._block_chain_bwd(chain, h, ud, dz, red_d, None)
                    self._ready(p + ".conv1.weight")
                    return
                if self.fuse_tail_bn and not getattr(ut, "gn", False):
                    # the reduce passes of the block's last BatchNorm and of the downsample BatchNorm ride on the tail's backward pass
                    st_t = self.bn[ut.bn]
                    bns = [(ut.y, st_t.mean, st_t.invstd)]
                    if ud is not None:
                        bns.append((ud.y, self.bn[ud.bn].mean, self.bn[ud.bn].invstd))
                    n_ = ops.bn_bwd_rows(out.B * out.H * out.W) * out.C
                    pz = [self._scratch(f"tz{i}", n_) for i in range(3)]
                    rows_t = ops.relu_bwd_sum_bn(o.v, o.grads, dz, bns, pz[0], pz[1:1 + len(bns)])
                    red_t = (pz[0], pz[1], rows_t)
                    red_d = (pz[0], pz[2], rows_t) if ud is not None else None
                else:
                    ops.relu_bwd_sum(o.v, o.grads, dz)
                dy = self._unit_bwd(ut, dz, 2, reduced=red_t)
                fz = None
                if self.kind == "basic" and self._fusable(ut, chain[-1][0]):
                    gx, half, fz = self._dgrad(ut, dy, p + ".g_t", fuse_into=chain[-1][0])
                    last_in.grads.append((gx, half))
                else:
                    last_in.grads.append(self._dgrad(ut, dy, p + ".g_t"))
                self._block_chain_bwd(chain, h, ud, dz, red_d, fz)
                self._ready(p + ".conv1.weight")
            tape.append(bwd)
        return o

    # ------------------------------------------------------------------------------------------------ BatchNorm from the Gram matrix
    def _gram_ok(self, p, a: View, out: View, bn=".bn3"):
        """does the block's conv3 -> bn3 -> (+ shortcut) -> ReLU run without a stored conv3 output?  bf16 bottlenecks whose conv3 input is
        64 ... gram_max_m wide; not in exact data-parallel mode (its exchanges are written against per-layer moments) and not when the block
        output also leaves as an fp8 image (the tail pass writes that image)"""
        # (the O(C m^2) finalizes cost the same whatever the map size: at 512 channels they pay on the large maps of the dilated DeepLab
        #  encoders - output stride 8: 64^2 pixels x 32 tiles - and lose on the 16^2 maps of the U-Net encoder)
        m_ok = 64 <= a.C <= self.gram_max_m or (a.C == 512 and self.gram_max_m >= 256 and a.B * a.H * a.W >= self.gram_512_pixels)
        return (self.gram_bn and self.kind == "bottleneck" and self.dtype == torch.bfloat16 and self.sync_bn is None and
                (p + bn + ".running_mean") in self.Bf and a.C % 64 == 0 and m_ok and out.C % 128 == 0 and self._q8_of(out) is None)

    def _unit_gram(self, a: View, conv, bn) -> Unit:
        """statistics of bn(conv1x1(a)) from the Gram matrix of a: scale / shift / saved statistics / running statistics, Q = W G kept for
        the backward.  The conv itself runs in _block with the BatchNorm in its epilogue."""
        wf = self.packed[conv]["wf"]
        C_, m = wf.shape[1], wf.shape[2]
        M = a.B * a.H * a.W
        st = self.bn[bn]
        G = self._buf(conv + ".gram", (m, m), torch.float64)
        sv = self._buf(conv + ".gram_s", (m,), torch.float64)
        q = self._buf(conv + ".gram_q", (C_, m), torch.float32)
        ops.SCOPE = "enc"
        ops.gram(a, G, sv, self._scratch("gram_ws", ops.gram_workspace_floats(M, m)))
        ops.bn_gram_finalize(G, sv, wf, M, self.P[bn + ".weight"], self.P[bn + ".bias"], self.Bf[bn + ".running_mean"], self.Bf[bn + ".running_var"],
                             st.scale, st.shift, st.mean, st.invstd, q)
        u = Unit(a, None, conv, bn, 1, 1, 0)
        u.gram, u.gram_s, u.gram_q = True, sv, q
        self.units[conv] = (u, None)
        return u

    def _gram_bwd(self, u: Unit, dz: View, name) -> View:
        """backward of conv1x1 -> BatchNorm given dz (gradient w.r.t. the BatchNorm output): R = dz^T a and sum dz from the weight-gradient
        GEMM, coefficients + dW + the packed two-source weight from cvcs_bn_gram_bwd / _mmat, the data gradient as ONE GEMM over [dz | a]"""
        a, st = u.x, self.bn[u.bn]
        wf = self.packed[u.conv]["wf"]
        C_, m = wf.shape[1], wf.shape[2]
        M = a.B * a.H * a.W
        R = self._scratch("gram_R", C_ * m)
        sdz = self._scratch("gram_sdz", C_)
        coef = self._scratch("gram_coef", 2 * C_)
        wd2 = self._buf(u.conv + ".wd2", (1, m, C_ + m))
        bias2 = self._buf(u.conv + ".bias2", (m,), torch.float32)
        need = ops.wgrad_workspace_floats_for(a, dz, 1, 1, 1, 0, with_bias=True)
        ops.conv2d_wgrad(a, dz, R, 1, 1, 1, 0, self._scratch("wg_ws", need), dbias=sdz)
        ops.bn_gram_bwd(R, sdz, wf, u.gram_q, u.gram_s, M, st.scale, st.mean, st.invstd, self.G[u.bn + ".weight"], self.G[u.bn + ".bias"],
                        self.G[u.conv + ".weight"], wd2, coef)
        ops.bn_gram_mmat(wf, coef, wd2, bias2, self._scratch("gram_mm", ops.bn_gram_mmat_workspace_floats(C_, m)))
        gx = ops.view(self._act(name, a.B, a.H, a.W, m))
        ops.SCOPE = "enc"
        ops.conv2d(dz, wd2, bias2, gx, 1, 1, x2=a)
        if self.keep_all:
            self.bwd_gram[u.conv] = dict(unit=u, dz=dz, gx=gx)
        return gx

    def _block_chain_bwd(self, chain, h: Act, ud: Unit | None, dz: View, red_d, fz):
        """backward of a block's conv chain (deepest unit first) and of its shortcut, given dz (the gradient behind the tail's ReLU): BatchNorm
        backward + weight gradient + data gradient of every unit; the gradients of the block input h are appended to h.grads - or, where
        _tail_fusable says so, summed and masked by the first conv's data-gradient launch itself (h.dz_ready)"""
        u1 = chain[0][0]
        gxd = None
        identity = ud is None      # the shortcut hands dz itself to the block input
        if ud is not None and getattr(ud, "gram", False):
            # same-resolution projection from the Gram path: weight gradient + ONE two-source data gradient, no BatchNorm passes
            h.grads.append((self._gram_bwd(ud, dz, ud.conv + ".gx"), False))
            ud, identity = None, False
        if ud is not None and self._tail_fusable(h, u1, ds=True):
            # (before the chain: the BatchNorm backward of the chain's units reuses the dy scratch this one writes)
            dyd = self._unit_bwd(ud, dz, 2, reduced=red_d)
            gxd = self._dgrad(ud, dyd, ud.conv + ".gx")
        tail_done = False
        for n_, (u, xin, aout) in reversed(list(enumerate(chain))):
            assert len(aout.grads) == 1
            dy = self._unit_bwd(u, aout.grads[0][0], 0, fused=fz)
            fz = None
            if n_ > 0 and self._fusable(u, chain[n_ - 1][0]):
                gx, half, fz = self._dgrad(u, dy, u.conv + ".gx", fuse_into=chain[n_ - 1][0])
                xin.grads.append((gx, half))
            elif n_ == 0 and identity and self._tail_fusable(h, u):
                self._dgrad_into_tail(u, dy, h, dz)
                tail_done = True
            elif n_ == 0 and gxd is not None:
                self._dgrad_into_tail(u, dy, h, h.grads[0][0] if h.grads else None, gxd)
                tail_done = True
            else:
                xin.grads.append(self._dgrad(u, dy, u.conv + ".gx"))
        if ud is not None and gxd is None:
            dyd = self._unit_bwd(ud, dz, 2, reduced=red_d)
            h.grads.append(self._dgrad(ud, dyd, ud.conv + ".gx"))
        elif identity and not tail_done:
            h.grads.append((dz, False))

    def _tail_fusable(self, h: Act, u1: Unit, ds: bool = False):
        """can the data-gradient launch of a block's first 1x1 conv write the PREVIOUS block's dz = (sum of the gradients of its output) *
        (out > 0) itself?  The previous block must be a Gram tail (its backward needs no BatchNorm reduce over dz), bf16, a channel count the
        1x1 taps kernel tiles.  Identity consumer (ds False): dz = (g_conv1 + this block's dz) * mask, the output may have no other consumer.
        Down-sampling consumer (ds True): dz = (g_conv1 + g_projection [half resolution] + the decoder's skip gradient) * mask."""
        ok = (self.fuse_tail_dz and getattr(h, "tail_dz", None) is not None and u1.k == 1 and u1.stride == 1 and
              self.dtype == torch.bfloat16 and h.v.C % 128 == 0 and not getattr(u1, "half_out", False))
        if not ok:
            return False
        if not ds:
            return not h.grads
        return len(h.grads) <= 1 and not any(half for _, half in h.grads) and h.v.H % 2 == 0 and h.v.W % 2 == 0

    def _dgrad_into_tail(self, u1: Unit, dy: View, h: Act, res: View | None, res2=None):
        v = h.v
        dzk = ops.view(self._act(h.tail_dz, v.B, v.H, v.W, v.C))
        ops.SCOPE = "enc"
        mb = getattr(h, "mask_bits", None)
        ops.conv2d(dy, self.packed[u1.conv]["wd"], None, dzk, 1, 1, res=res, res2=res2, mask=v if mb is None else None, mask_bits=mb)
        h.dz_ready = dzk
        h.grads = []
        if self.keep_all:
            self.bwd_units[u1.conv]["gx_tail"] = (dzk, res, res2, v)

    # ------------------------------------------------------------------------------------------------ backward pieces
    def _fusable(self, producer: Unit, consumer: Unit):
        """can the data-gradient launch of `producer` (whose output is the gradient of `consumer`'s activation) also take the
        reduce pass of consumer's BatchNorm backward?  (cvcs_conv_desc.bwd_y: bf16 3x3 / stride 1 launches on the three-
        workgroup narrow tiles, i.e. up to 128 channels - on the wide kernel the longer epilogue is exposed)"""
        y = consumer.y
        if getattr(producer, "fp8", False):      # the fp8 kernel has no fused reduce epilogue
            return False
        if not (self.fuse_bn_bwd and not getattr(consumer, "gn", False) and self.dtype == torch.bfloat16 and producer.k == 3 and producer.stride == 1 and
                producer.dil == 1 and y.H >= 8 and y.W >= 8):
            return False
        if self.thin_bn_bwd and y.C in (16, 32) and producer.y is not None and producer.y.C in (16, 32) and y.W >= 16:
            return True       # the thin kernel of the light decoder stages (one partial row per workgroup)
        return y.C <= 128 and y.C % 64 == 0

    def _unit_bwd(self, u: Unit, g: View, mode, fused=None, reduced=None) -> View:
        """BatchNorm (+ReLU, mode 0) backward of a unit, its conv's weight gradient; returns dy (gradient w.r.t. the conv
        output) in a scratch buffer that lives until the next _unit_bwd.  fused = (part_dz, part_dzx, rows): the reduce pass
        was taken by the launch that produced g (one partial row per 16x16 tile)"""
        y = u.y
        M, C_ = y.B * y.H * y.W, y.C
        if getattr(u, "gn", False):
            st, HW = self.gn[u.bn], y.H * y.W
            part = self._scratch("gn_part", y.B * ops.gn_rows(HW) * 2 * C_)
            ops.gn_act_bwd_reduce(y, g, st["scale"], st["shift"], True, part)
            ops.gn_bwd_finalize(part, y.B, HW, C_, GN_GROUPS, self.P[u.bn + ".weight"], st["mean"], st["invstd"], st["tmp"],
                                self.G[u.bn + ".weight"], self.G[u.bn + ".bias"], st["ca"], st["cb"], st["cc"])
            dy = ops.view(self._scratch(u.conv + ".dy" if self.keep_all else "dy", M * C_, self.dtype).view(y.B, y.H, y.W, C_))
            ops.gn_act_bwd_apply(y, g, st["scale"], st["shift"], True, st["ca"], st["cb"], st["cc"], dy)
            need = ops.wgrad_workspace_floats_for(u.x, dy, u.k, u.k, u.stride, u.pad, dil=u.dil)
            ops.conv2d_wgrad(u.x, dy, self.G[u.conv + ".weight"], u.k, u.k, u.stride, u.pad, self._scratch("wg_ws", need), dil=u.dil)
            return dy
        st = self.bn[u.bn]
        rows = ops.bn_bwd_rows(M)
        p0, p1, p2 = (self._scratch(f"bnb{i}", rows * C_) for i in range(3))
        if reduced is not None:
            # reduced = (part_dz, part_dzx, rows): the reduce pass rode on the block-tail backward (ops.relu_bwd_sum_bn), rows as bn_bwd_rows
            r0, r1, rrows = reduced
            ops.bn_bwd_finalize(r0, r1, rrows, M, C_, self.P[u.bn + ".weight"], st.invstd, self.G[u.bn + ".weight"], self.G[u.bn + ".bias"],
                                st.ca, st.cb)
        elif fused is not None:
            f0, f1, trows = fused
            if trows <= 4096:       # the finalize takes the tile rows as they are (two column-sum launches less per unit)
                ops.bn_bwd_finalize(f0, f1, trows, M, C_, self.P[u.bn + ".weight"], st.invstd, self.G[u.bn + ".weight"],
                                    self.G[u.bn + ".bias"], st.ca, st.cb)
            else:
                rows1 = ops.bn_bwd_rows(trows)
                for src, dst in ((f0, p0), (f1, p1)):
                    ops.colsum_partial(View(src[:trows * C_].view(1, trows, 1, C_), 0, C_), dst[:rows1 * C_])
                ops.bn_bwd_finalize(p0, p1, rows1, M, C_, self.P[u.bn + ".weight"], st.invstd, self.G[u.bn + ".weight"],
                                    self.G[u.bn + ".bias"], st.ca, st.cb)
        else:
            ops.bn_bwd_reduce(y, g, None, st.scale, st.shift, st.mean, st.invstd, mode, p0, p1)
            ops.bn_bwd_finalize(p0, p1, rows, M, C_, self.P[u.bn + ".weight"], st.invstd, self.G[u.bn + ".weight"], self.G[u.bn + ".bias"],
                                st.ca, st.cb)
        if self.sync_bn is not None:
            # dgamma / dbeta stay this rank's sums (the gradient all-reduce adds them up); the two coefficients of the apply pass are
            # means over every rank's pixels
            sums = self._sync_sums[:2 * C_]
            gb, gw = self.G[u.bn + ".bias"], self.G[u.bn + ".weight"]

            def exchange(sums=sums, gb=gb, gw=gw, n=C_):
                sums[:n].copy_(gb)
                sums[n:].copy_(gw)
                self.sync_bn.all_reduce(sums)
            self._host(exchange)
            ops.bn_bwd_coeffs(sums, M * self.sync_bn.world, C_, st.ca, st.cb)
        # two dy buffers in turn: the weight gradient of unit k (side lane) may still read its dy while unit k + 1 writes the other one
        self._dy_toggle ^= 1
        dyname = u.conv + ".dy" if self.keep_all else ("dy" if not self.overlap_wgrad else f"dy{self._dy_toggle}")
        dy = ops.view(self._scratch(dyname, M * C_, self.dtype).view(y.B, y.H, y.W, C_))
        if self.keep_all:
            self.bwd_units[u.conv] = dict(unit=u, g=g, mode=mode, dy=dy)
        rec = _lib._recording
        if self.overlap_wgrad and rec is not None and dyname in self._dy_reader:
            rec.wait_side(self._dy_reader[dyname])
        q8 = None
        if getattr(u, "fp8", False):      # the data gradient of this unit runs fp8: dy leaves the pass with its e5m2 image
            self._register_q8(dy.t, "g:" + u.conv, _lib.E5M2, buf=None if self.keep_all else "dy")
            q8 = self._q8_of(dy)
        ops.bn_bwd_apply(y, g, None, st.scale, st.shift, st.mean, st.invstd, st.ca, st.cb, mode, dy, p2, q8=q8)
        side = self.overlap_wgrad and rec is not None
        ws_name = "wg_ws_side" if side else "wg_ws"      # (the side lane's launches share one workspace: their stream serialises them)
        with (_lib.side_lane() if side else contextlib.nullcontext()):
            self._wgrad(u, dy, ws_name)
        if side:
            self._dy_reader[dyname] = rec.last_index()
        return dy

    def _wgrad(self, u: Unit, dy: View, ws_name):
        """weight gradient of a unit's convolution into self.G (an engine with other convolution kinds overrides this)"""
        if u.virt:
            need = ops.wgrad_workspace_floats_for(u.x, dy, 7, 1, 2, 3, virt=True)
            ops.conv2d_wgrad(u.x, dy, self.stem_dw_tmp, 7, 1, 2, 3, self._scratch(ws_name, need), cin_real=32, virt=True)
            ops.unpack_stem_wgrad(self.stem_dw_tmp, self.G[u.conv + ".weight"])
        else:
            need = ops.wgrad_workspace_floats_for(u.x, dy, u.k, u.k, u.stride, u.pad, dil=u.dil)
            ops.conv2d_wgrad(u.x, dy, self.G[u.conv + ".weight"], u.k, u.k, u.stride, u.pad, self._scratch(ws_name, need), dil=u.dil)

    def _dgrad(self, u: Unit, dy: View, name, fuse_into: Unit | None = None):
        """data gradient of a unit's conv -> (view, half_resolution).  fuse_into: the unit whose activation this gradient
        belongs to (mode-0 BatchNorm backward): its reduce pass rides on this launch; returns (view, half, fused partials)"""
        if fuse_into is not None:
            yv, st = fuse_into.y, self.bn[fuse_into.bn]
            trows = ops.conv_stat_rows(dy, yv.C, 3, 3, 1, 1, bwd=True)
            f0, f1 = self._scratch("fz0", trows * yv.C), self._scratch("fz1", trows * yv.C)
            gx = ops.view(self._act(name, yv.B, yv.H, yv.W, yv.C))
            ops.SCOPE = "enc" if u.conv.startswith("encoder.") else "dec"
            ops.conv2d(dy, self.packed[u.conv]["wd"], None, gx, 3, 3, 1, 1, bn_bwd=(yv, st.scale, st.shift, st.mean, st.invstd, 0, f0, f1))
            if self.keep_all:
                self.bwd_units[u.conv]["gx"] = gx
            return gx, False, (f0, f1, trows)
        wd = self.packed[u.conv]["wd"]
        x = u.x
        B, cin = x.B, x.C
        ops.SCOPE = "enc" if u.conv.startswith("encoder.") else "dec"
        if u.k == 1:
            # stride 2: the gradient exists at the even pixels only - computed at half resolution, scattered by the consumer
            gx = ops.view(self._act(name, B, dy.H, dy.W, cin))
            ops.conv2d(dy, wd, None, gx, 1, 1)
            if self.keep_all:
                self.bwd_units[u.conv]["gx"] = gx
            return gx, (u.stride == 2 or getattr(u, "half_out", False))
        gx = ops.view(self._act(name, B, x.H, x.W, cin))
        if getattr(u, "fp8", False):
            # e5m2 gradient (range over precision) x e4m3 flipped weights on the same kernel
            dy8, ig = self._fp8_input(dy, "g:" + u.conv, _lib.E5M2, True, buf="dy" if not self.keep_all else None)
            ops.conv3x3_fp8(dy8, _lib.E5M2, self.packed[u.conv]["wd8"], gx, self.f8, ig, self.f8.slot("w:" + u.conv, _lib.E4M3))
        elif u.stride == 1:
            ops.conv2d(dy, wd, None, gx, 3, 3, 1, u.dil, u.dil)   # (dilated: the generic gather kernel, see wgrad_dilated)
        else:
            # 3x3 / stride 2 / pad 1: dx = conv3x3/s1/p1(zero-dilated dy, flipped weights) - on the halo kernel
            dil = ops.view(self._scratch("dilated", B * x.H * x.W * dy.C, self.dtype).view(B, x.H, x.W, dy.C))
            ops.dilate2x(dy, dil)
            ops.conv2d(dil, wd, None, gx, 3, 3, 1, 1, flops=2.0 * B * dy.H * dy.W * dy.C * cin * 9)
        if self.keep_all:
            self.bwd_units[u.conv]["gx"] = gx
        return gx, False

    def _host(self, fn):
        """run fn() now and at this point of every replay of the plan being recorded (eager pass outside a recording: just now)"""
        if _lib._recording is not None:
            _lib._recording.host(fn)
        else:
            fn()

    def _ready(self, name):
        off = self._goff[name]   # (the hook is looked up at replay time: a data-parallel wrapper may be attached later)
        _lib._recording.host(lambda: self.on_grad_ready(off, []) if self.on_grad_ready is not None else None)

    # ------------------------------------------------------------------------------------------------ the network
    def _backbone(self, B, S, train):
        """issues every launch between the packed input and the last decoder activation; returns that activation and, in
        train mode, fills self._tape with the backward closures (run in reverse by backward())"""
        tape = []
        self.units, self.relu_order = {}, []     # relu_order: every ReLU output in execution order (tests)
        self._q8, self._train_pass = {}, train
        self.refresh_weights(train)
        dec, widths = self.dec, self.widths
        skipc = [widths[2], widths[1], widths[0], 64, 0]
        # decoder input buffers: [up-sampled | skip] channel ranges; the encoder writes its features into the skip ranges
        cat, cin = [], widths[3]
        for i in range(5):
            s = S >> (4 - i)
            cat.append(self._act(f"cat{i}", B, s, s, cin + skipc[i]))
            if train and self._fp8_ok(f"decoder.blocks.{i}.conv1.0", ops.view(cat[i]), 3, 1, 1):
                self._register_q8(cat[i], f"a:decoder.blocks.{i}.conv1.0")      # both producers (up-sampling, encoder block tail) write into it
            cin = dec[i]
        upc = [widths[3]] + list(dec[:4])                    # channels of the up-sampled part of cat[i]
        skip_view = lambda i: View(cat[i], upc[i], skipc[i])  # noqa: E731
        f1, p0 = self._stem(B, S, train, tape, skip_view(3))
        feats = self._stages(p0, train, tape, lambda s: skip_view(3 - s) if s < 4 else None)   # stage s -> cat[3 - s]
        h = feats[4]
        # decoder
        for i in range(5):
            s = S >> (4 - i)
            up = View(cat[i], 0, upc[i])
            # (opt-in, CVCS_FUSE_UPSAMPLE=1) a stage without a skip feature whose first conv is the thin 32 -> 16 instance - the full-resolution stage
            # of the default decoder - never stores its up-sampled input: forward and weight gradient interpolate while they stage (ops.UpView)
            fuse_up = (self.fuse_upsample and train and skipc[i] == 0 and self.dtype == torch.bfloat16 and upc[i] == 32 and dec[i] == 16 and
                       not self.keep_all and self._q8_of(up) is None)
            if not fuse_up:
                ops.upsample2x_fwd(h.v, up, q8=self._q8_of(up))
            pre = f"decoder.blocks.{i}"
            a1 = Act(ops.view(self._act(pre + ".a1", B, s, s, dec[i])))
            a2 = Act(ops.view(self._act(pre + ".a2", B, s, s, dec[i])))
            if train and self._fp8_ok(pre + ".conv2.0", a1.v, 3, 1, 1):
                self._register_q8(a1.v.t, "a:" + pre + ".conv2.0")
            xin = ops.up_view(h.v) if fuse_up else ops.view(cat[i])
            u1 = self._unit(xin, pre + ".conv1.0", pre + ".conv1.1", 3, 1, 1, train, a1.v)
            u2 = self._unit(a1.v, pre + ".conv2.0", pre + ".conv2.1", 3, 1, 1, train, a2.v)
            if train:
                def dec_bwd(i=i, s=s, pre=pre, u1=u1, u2=u2, a1=a1, a2=a2, prev=h):
                    assert len(a2.grads) == 1
                    dy = self._unit_bwd(u2, a2.grads[0][0], 0)
                    if self._fusable(u2, u1):
                        g1, _, fz = self._dgrad(u2, dy, pre + ".g_a1", fuse_into=u1)
                    else:
                        (g1, _), fz = self._dgrad(u2, dy, pre + ".g_a1"), None
                    dy = self._unit_bwd(u1, g1, 0, fused=fz)
                    gcat, _ = self._dgrad(u1, dy, pre + ".g_cat")
                    gprev = ops.view(self._act(pre + ".g_prev", B, s // 2, s // 2, upc[i]))
                    ops.upsample2x_bwd(View(gcat.t, 0, upc[i]), gprev)
                    prev.grads.append((gprev, False))
                    if skipc[i]:
                        skip_act = f1 if i == 3 else feats[3 - i]
                        skip_act.grads.append((View(gcat.t, upc[i], skipc[i]), False))
                    self._ready(pre + ".conv1.0.weight")
                tape.append(dec_bwd)
            h = a2
        if train:
            self._tape = tape
        self._last = h
        return h

    def _stages(self, h: Act, train, tape, stage_out, output_stride=32):
        """the four residual stages after the stem's max-pool; stage_out(s) -> view the last block of stage s writes its
        output to (a channel range of a decoder's concat buffer) or None; returns {stage: Act}"""
        feats, B = {}, h.v.B
        grid = 1      # sub-grid layout L_grid of h (cvcs_regrid): a block of dilation d runs as a PLAIN 3x3 block on the B*d*d sub-sampled
        #               images of L_d - halo kernels for forward, data and weight gradient instead of the generic gather kernel and nine
        #               shifted 1x1 weight-gradient problems; 1x1 convs, BatchNorm, ReLU and the residual add do not care about pixel order
        plan = stage_plan(self.arch, output_stride)
        for i, (s, b, stride, dil) in enumerate(plan):
            p = f"encoder.layer{s}.{b}"
            hfull = h.v.H * grid
            hs, w = hfull // stride, self.widths[s - 1]
            last = b == self.depths[s - 1] - 1
            out = stage_out(s) if last else None
            want = dil if (self.grid_dilated and dil > 1 and stride == 1 and out is None and hs % dil == 0) else 1
            if want != grid:
                h = self._regrid(h, B, hfull, grid, want, train, tape, p + ".regrid")
                grid = want
            if grid > 1:
                out = ops.view(self._act(p + ".out", B * grid * grid, hs // grid, hs // grid, w))
                self._grid = grid
                h = self._block(h, p, 1, out, train, tape, 1)
                self._grid = 1
                if i + 1 == len(plan):      # the consumers of the last stage read the ordinary layout
                    h = self._regrid(h, B, hs, grid, 1, train, tape, p + ".ungrid")
                    grid = 1
            else:
                if out is None:
                    out = ops.view(self._act(p + ".out", B, hs, hs, w))
                h = self._block(h, p, stride, out, train, tape, dil)
            if grid == 1:
                feats[s] = h
        return feats

    def _regrid(self, h: Act, B, H, a, b, train, tape, name) -> Act:
        """h in sub-grid layout L_a of a [B, H, H, C] map -> a new activation in L_b; the gradient is the same gather from L_b to L_a"""
        C_ = h.v.C
        out = Act(ops.view(self._act(name, B * b * b, H // b, H // b, C_)))
        ops.regrid(h.v, out.v, H, H, a, b)
        if train:
            def bwd():
                assert 1 <= len(out.grads) <= 3 and not any(half for _, half in out.grads)
                g = out.grads[0][0]
                if len(out.grads) > 1:
                    g = ops.view(self._act(name + ".gsum", B * b * b, H // b, H // b, C_))
                    ops.relu_bwd_sum(None, out.grads, g)
                gx = ops.view(self._act(name + ".gx", B * a * a, H // a, H // a, C_))
                ops.regrid(g, gx, H, H, b, a)
                h.grads.append((gx, False))
            tape.append(bwd)
        return out

    def _stem(self, B, S, train, tape, f1_view: View | None = None):
        """packed input -> 7x7/s2 conv -> BN -> ReLU (f1, optionally into a concat buffer) -> 3x3/s2 max-pool (p0)"""
        in4 = ops.view(self._buf("in4", (B, S, S + 8, 4)))
        f1 = Act(f1_view if f1_view is not None else ops.view(self._act("stem.f1", B, S // 2, S // 2, 64)))
        u0 = self._unit(in4, "encoder.conv1", "encoder.bn1", 7, 2, 3, train, f1.v, virt=True)
        p0 = Act(ops.view(self._act("pool0", B, S // 4, S // 4, 64)))
        idx = self._buf("pool0.idx", (B * (S // 4) * (S // 4) * 64,), torch.uint8)
        ops.maxpool3x3s2_fwd(f1.v, p0.v, idx)
        if train:
            def stem_bwd():
                g = [v for v, _ in p0.grads]
                dz = ops.view(self._act("stem.dz", B, S // 2, S // 2, 64))
                if self.fuse_stem_bwd and len(g) <= 2 and len(f1.grads) <= 3 and not any(half for _, half in f1.grads) and self.sync_bn is None:
                    # max-pool backward + the skip gradients + the ReLU mask + the reduce pass of the stem's BatchNorm in ONE launch (four
                    # passes over the largest encoder map before: 0.66 ms of the 512^2 x 32 step)
                    st0 = self.bn[u0.bn]
                    n_ = ops.bn_bwd_rows(B * (S // 2) * (S // 2)) * 64
                    pz = [self._scratch(f"tz{i}", n_) for i in range(2)]
                    rows_t = ops.relu_bwd_sum_bn(f1.v, f1.grads, dz, [(u0.y, st0.mean, st0.invstd)], pz[0], pz[1:2], pooled=(g, idx))
                    self._unit_bwd(u0, dz, 0, reduced=(pz[0], pz[1], rows_t))
                else:
                    dx = ops.view(self._act("pool0.dx", B, S // 2, S // 2, 64))
                    ops.maxpool3x3s2_bwd(g[0], g[1] if len(g) > 1 else None, idx, dx)
                    f1.grads.append((dx, False))
                    ops.relu_bwd_sum(f1.v, f1.grads, dz)
                    self._unit_bwd(u0, dz, 0)
                self._ready("encoder.conv1.weight")
            tape.append(stem_bwd)
        return f1, p0

    # everything an input shape owns: recorded plans, activation buffers, the views the plans were recorded against
    SHAPE_STATE = ("_rec", "_bufs", "gn", "logits", "logits_lo", "dlogits_lo", "_last_act", "_tape", "_last", "units", "relu_order", "_q8",
                   "_saved_train", "_head_pending", "_head_done", "_head_rows", "hs_order")
    PLAN_CACHE = int(os.environ.get("CVCS_PLAN_CACHE", "3"))      # input shapes kept alive (least recently used one dropped)

    def _switch_shape(self, B, S) -> bool:
        """make (B, S) the current input shape.  The state of the shape being left - buffers and recorded launch plans - is parked and comes
        back when that shape returns (train -> eval_model / validation_loss -> train every epoch: no re-allocation, no re-recording; under N
        ranks no rank stalls the others re-planning).  -> True when (B, S) is new and its state has to be built"""
        st = self._shape_cache.pop((B, S), None)
        if self.shape is not None:
            self._shape_cache[self.shape] = {a: getattr(self, a) for a in self.SHAPE_STATE if hasattr(self, a)}
            self._shape_cache.move_to_end(self.shape)
            while len(self._shape_cache) > max(self.PLAN_CACHE - 1, 0):
                self._shape_cache.popitem(last=False)
        self.shape = (B, S)
        if st is None:
            self._last_act, self._head_pending, self._head_done = {}, False, False
            return True
        for a, v in st.items():
            setattr(self, a, v)
        return False

    def _plan(self, B, S):
        if self.shape != (B, S):
            assert S % 32 == 0, "tile side must be a multiple of 32 (the encoder reduces the resolution 32 times)"
            if self._switch_shape(B, S):
                self._rec, self._bufs, self.gn = {}, {}, {}
                self.logits = torch.empty((B, self.NC, S, S), dtype=torch.float32, device=self.dev)

    def _run(self, key, fn):
        """first time: run fn() eagerly while recording its launches; afterwards replay the recording"""
        rec = self._rec.get(key)
        if rec is None:
            self._dy_reader = {}
            with _lib.Recording() as rec:
                fn()
            self._rec[key] = rec
        elif self.overlap_wgrad and key == "bwd":
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=self.dev)
            rec.replay(torch.cuda.current_stream().cuda_stream, ops.TIMERS, side=self._side_stream)
        else:
            rec.replay(torch.cuda.current_stream().cuda_stream, ops.TIMERS)

    def _forward_backbone(self, x, train):
        B, C_, S, S2 = x.shape
        assert C_ == 3 and S == S2
        self._plan(B, S)
        ops.pack_input_stem(x.contiguous(), self._buf("in4", (B, S, S + 8, 4)))
        key = "fwd_train" if train else "fwd_eval"
        if train and "fwd_train" not in self._rec:
            self._rec.pop("bwd", None)
        self._run(key, lambda: self._backbone(B, S, train))
        if key not in self._last_act:
            self._last_act[key] = self._last
        self._saved_train = train
        return self._last_act[key]

    def forward(self, x: torch.Tensor, train: bool, lazy: bool = False) -> torch.Tensor:
        """x: NCHW u8 | f32 [B,3,S,S] on the device -> NCHW f32 logits [B,NC,S,S] (owned by the engine).
        lazy (train mode): the head is NOT run - the returned buffer is filled by materialize_logits() if anybody reads the logits; a loss that
        recognises the handle (utils.CrossEntropyLoss) calls head_ce() instead and the logits never exist (nets.LazyLogits)"""
        h = self._forward_backbone(x, train)
        self.fwd_serial += 1
        self._head_done = False
        self._head_pending = bool(lazy and train and self.lazy_head_ok())
        if not self._head_pending:
            ops.head_fwd(h.v, self.P["segmentation_head.0.weight"].view(self.NC, self.dec[-1]), self.P["segmentation_head.0.bias"], self.logits)
        return self.logits

    def lazy_head_ok(self) -> bool:
        """(the engines derived from this one - DeepLab, Swin + UPerNet, Segformer, MobileNet - have other heads: their logits stay eager)"""
        return type(self) is ResNetUNetEngine and self.dtype == torch.bfloat16 and self.dec[-1] in (16, 32, 64) and self.NC <= 32

    def materialize_logits(self):
        if self._head_pending:
            h = self._last_act["fwd_train"]
            ops.head_fwd(h.v, self.P["segmentation_head.0.weight"].view(self.NC, self.dec[-1]), self.P["segmentation_head.0.bias"], self.logits)
            self._head_pending = False

    def head_ce(self, target, class_weight, ignore_index, workspace, loss_out):
        """head + cross-entropy + the head's backward in one pass over the last decoder activation (cvcs_head_ce): the gradient w.r.t. that
        activation and the partial dW | db rows are kept for backward(), which then needs no logit gradient"""
        B, S = self.shape
        h = self._last_act["fwd_train"]
        NC, CH = self.NC, self.dec[-1]
        rows = ops.head_ce_rows(B * S * S)
        part = self._scratch("head_part", rows * (NC * CH + NC))
        gh = ops.view(self._act("head.gx", B, S, S, CH))
        ops.head_ce(h.v, self.P["segmentation_head.0.weight"].view(NC, CH), self.P["segmentation_head.0.bias"], target, class_weight, ignore_index,
                    1.0, gh, part, workspace, loss_out)
        self._head_done, self._head_rows = True, rows

    def scale_head_grads(self, gout):
        """the loss was scaled / combined by further autograd ops: its incoming gradient (a device scalar) multiplies what head_ce produced"""
        B, S = self.shape
        NC, CH = self.NC, self.dec[-1]
        g = gout.detach().to(device=self.dev, dtype=torch.float32).reshape(1).contiguous()
        ops.scale_unless_one(self._act("head.gx", B, S, S, CH), g)
        ops.scale_unless_one(self._scratch("head_part", self._head_rows * (NC * CH + NC)), g)

    def forward_labels(self, x: torch.Tensor, labels: torch.Tensor):
        h = self._forward_backbone(x, False)
        ops.head_argmax(h.v, self.P["segmentation_head.0.weight"].view(self.NC, self.dec[-1]), self.P["segmentation_head.0.bias"], labels)
        return labels

    def backward(self, dlogits: torch.Tensor):
        """dlogits: NCHW f32 [B,NC,S,S]; fills every gradient view in self.G (overwrites)"""
        assert self._saved_train, "backward needs a preceding forward in train mode"
        if self.on_backward_begin is not None:
            self.on_backward_begin()
        B, S = self.shape
        NC = self.NC
        h = self._last_act["fwd_train"]
        CH = self.dec[-1]
        gh = ops.view(self._act("head.gx", B, S, S, CH))
        if self._head_done:       # head_ce() already produced gh and the partial rows
            rows = self._head_rows
            part = self._scratch("head_part", rows * (NC * CH + NC))
            self._head_done = False
        else:
            assert not self._head_pending or dlogits is not None
            rows = ops.head_bwd_rows(B * S * S)
            part = self._scratch("head_part", rows * (NC * CH + NC))
            ops.head_bwd(h.v, dlogits.contiguous(), self.P["segmentation_head.0.weight"].view(NC, CH), gh, part)
        gw, gb = self.G["segmentation_head.0.weight"], self.G["segmentation_head.0.bias"]
        assert gb.data_ptr() == gw.data_ptr() + gw.numel() * 4
        ops.colsum_finalize(part, rows, NC * CH + NC, gw)
        if self.on_grad_ready is not None:
            self.on_grad_ready(self._goff["segmentation_head.0.weight"], [])

        def run_tape():
            h.grads = [(gh, False)]
            for fn in reversed(self._tape):
                fn()
            if self.fp8:
                self.f8.update()      # delayed scaling: this step's amax of every quantised tensor becomes the next step's scale
        self._run("bwd", run_tape)
