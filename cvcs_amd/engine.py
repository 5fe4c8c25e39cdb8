"""Execution plan of the reference's two U-Nets on the HIP kernels.

`UNetEngine` owns every device buffer of one (batch, tile-size) shape - NHWC activations, their gradients,
BatchNorm statistics, packed weights - and issues the forward / backward launch sequence through the C-ABI.
The network wiring follows nets.Urnet.forward (source/scripts/nets.py:93-115) and nets.Urnetv2.forward
(:176-199); the block order follows blocks.UnetEncodeLayer (conv -> BN -> ReLU, blocks.py:8-23) and
blocks.UnetForwardDecodeLayer (conv -> ReLU -> BN twice, blocks.py:36-49).

MI355X-first choices (none of them a translation of the reference's op sequence):
  * torch.concat never runs: the skip tensor and the up-sampled tensor are written by their producers into the two
    channel halves of one NHWC buffer, and the decoder conv reads it as a single 2C-channel input;
  * MaxPool is written by the same pass that applies BN+ReLU; its backward is folded into the BN backward passes
    (the arg-max is recomputed from the saved pre-BN tensor, no index tensor is stored);
  * BatchNorm batch statistics come out of the conv epilogue; BN backward is two streaming passes;
  * ConvTranspose2d(k2,s2) is a 1x1 implicit GEMM with a pixel-shuffling epilogue straight into the concat buffer.
"""
from __future__ import annotations

import os

import torch

from . import _lib, ops
from .ops import View

WIDTHS = (64, 128, 256, 512, 1024)
# concat order per decoder stage d=1..4: True -> (skip, up) ; False -> (up, skip)   (nets.py:185,189,193,197)
SKIP_FIRST = {1: True, 2: True, 3: False, 4: True}


class _BN:
    """per-BatchNorm device state: parameters (views of the flat buffers) + per-step vectors."""

    def __init__(self, C, dev):
        f = lambda: torch.empty(C, dtype=torch.float32, device=dev)  # noqa: E731
        self.scale, self.shift, self.mean, self.invstd, self.ca, self.cb = f(), f(), f(), f(), f(), f()


class UNetEngine:
    def __init__(self, variant: str, num_classes: int, dtype: torch.dtype, device, fp8: bool = False):
        assert variant in ("Unet", "Unetv2")
        assert not fp8 or dtype == torch.bfloat16, "the fp8 convolutions live beside bf16 storage (precision 'fp8' = mixed bf16 / fp8)"
        self.variant, self.NC, self.dtype, self.dev = variant, num_classes, dtype, torch.device(device)
        # precision "fp8" (BASELINE configs[4]'s "mixed bf16/fp8 convs" on the reference's own U-Nets): the 3x3 convolutions whose channel counts
        # are multiples of 128 (levels 2-5 and their decoder stages) run forward AND data gradient on the block-scaled fp8 MFMA
        # (csrc/conv_fp8.hip) in train mode; weight gradients, the 64-channel level and every evaluation pass stay bf16
        self.fp8 = bool(fp8)
        self.f8 = None
        self._q8buf = {}
        self._produced = set()     # bf16 tensors whose fp8 image was written by their producer pass in this step
        self.code = ops.dtype_code(dtype)
        self.KG = ops.KGROUP[self.code]
        self.shape = None
        self.P = None   # name -> f32 parameter tensor (views of the module's flat buffer)
        self.G = None   # name -> f32 gradient tensor (views of the flat gradient buffer)
        self.Bf = None  # name -> f32 buffer (running stats)
        self.packed = {}
        self._saved_train = False
        self.fwd_serial = 0              # forward passes so far (a nets.LazyLogits handle is current while it matches)
        self._head_pending = self._head_done = False
        self._head_rows, self._head_in, self._head_w = 0, None, None
        self._pack_table = None   # device table of (master weight, packed operands) built at first use
        self.sync_bn = None             # parallel.SyncStats: batch-norm statistics over all ranks' tiles (None: per rank)
        self._sync_mom = self._sync_sums = None   # its exchange buffers (allocated by enable_sync_bn)
        self.head_wf = torch.empty((num_classes, 64), dtype=torch.float32, device=self.dev)   # head with the last BN folded in
        self.head_bf = torch.empty(num_classes, dtype=torch.float32, device=self.dev)
        self.one = torch.ones(1024, dtype=torch.float32, device=self.dev)    # identity affine of the pooling-only pass
        self.zero = torch.zeros(1024, dtype=torch.float32, device=self.dev)
        self.on_backward_begin = None   # data-parallel hooks (cvcs_amd.parallel): called at the start of backward,
        self.on_grad_ready = None       # and with the lowest flat-gradient offset that is complete so far
        # Weight gradients are off the critical chain of backward (dy_L -> dgrad_L -> BN backward_{L-1} -> ...): they can be
        # issued on a second HIP stream, so that the MFMA-bound wgrad kernels co-run with the HBM-bound BatchNorm / pooling
        # passes of the layers below.  Measured +2.5 % tiles/s at B=32, 512x512 (the two MFMA-bound kernel families
        # mostly just share the matrix pipes); off by default because it blurs per-kernel timing attribution
        # (CVCS_OVERLAP_WGRAD=1 switches it on).
        self.overlap_wgrad = os.environ.get("CVCS_OVERLAP_WGRAD", "0") == "1"
        # the data gradient of the conv above an enc_L.0 / dec_L.0 block also takes the first pass of that block's BatchNorm backward
        self.fuse_bn_bwd = os.environ.get("CVCS_FUSE_BN_BWD", "1") == "1"
        # ... where that launch runs the three-workgroup narrow kernel (64- and 128-channel blocks: +0.6 % on the step each); on the
        # wide kernel, one workgroup per CU, nothing hides the longer epilogue and the step got 0.6 % slower (tuning knobs)
        self.fuse_bn_bwd_c = (int(os.environ.get("CVCS_FUSE_BN_BWD_MINC", "0")), int(os.environ.get("CVCS_FUSE_BN_BWD_MAXC", "128")))
        self._side = None
        self._side_event = None

    # ------------------------------------------------------------------------------------------------ binding
    def bind(self, params, grads, buffers):
        self.P, self.G, self.Bf = params, grads, buffers
        base = min(g.data_ptr() for g in grads.values())
        self._goff = {k: (g.data_ptr() - base) // 4 for k, g in grads.items()}
        dev, dt = self.dev, self.dtype
        self.packed = {}
        self._pack_table = None
        for name, w in params.items():
            if w.dim() != 4 or name == "decode_forward4.1.weight":
                continue
            key = name[:-len(".weight")]
            if name.startswith("upscale") and self.variant == "Unetv2":
                cin, cout = w.shape[:2]
                self.packed[key] = dict(kind="convT", wf=torch.empty(1, 4 * cout, cin, dtype=dt, device=dev),
                                        b4=torch.empty(4 * cout, dtype=torch.float32, device=dev),
                                        wd=torch.empty(4, cin, cout, dtype=dt, device=dev))
            else:
                cout, cin, kh, kw = w.shape
                cpad = max(cin, self.KG) if cin % self.KG else cin
                first = cin == 3
                self.packed[key] = dict(kind="conv", wf=torch.empty(kh * kw, cout, cpad, dtype=dt, device=dev),
                                        wd=None if first else torch.empty(kh * kw, cin, cout, dtype=dt, device=dev))
                if self.fp8 and kh * kw == 9 and cout % 128 == 0 and cin % 128 == 0:
                    pk = self.packed[key]
                    pk["wf8"], pk["wd8"] = (torch.empty(t.shape, dtype=torch.uint8, device=dev) for t in (pk["wf"], pk["wd"]))
        if self.fp8:
            self.f8 = ops.Fp8Slots(dev)

    def refresh_weights(self, train=True):
        """re-pack the f32 master weights into the MFMA operand layouts (after every optimiser step): every 3x3 conv in one
        launch driven by a device-resident table, the four ConvTranspose layers by their own small kernels."""
        if self._pack_table is None:
            entries = [(self.P[k + ".weight"], pk["wf"], pk["wd"]) for k, pk in self.packed.items() if pk["kind"] == "conv"]
            self._pack_table = ops.pack_table(entries, self.dev)
        ops.pack_conv_weights(self._pack_table[0], self._pack_table[1], self.dtype)
        for key, pk in self.packed.items():
            if pk["kind"] == "convT":
                ops.pack_convT_weight_into(self.P[key + ".weight"], self.P[key + ".bias"], pk["wf"], pk["b4"], pk["wd"])
        if self.fp8 and train:
            # e4m3 images of the packed bf16 weights, one scale per layer (forward and data-gradient image hold the same values)
            for key, pk in self.packed.items():
                if "wf8" in pk:
                    i = self.f8.slot("w:" + key, _lib.E4M3)
                    for src, dst in ((pk["wf"], pk["wf8"]), (pk["wd"], pk["wd8"])):
                        rows, k = src.shape[0] * src.shape[1], src.shape[2]
                        ops.quantize_fp8(View(src.view(1, rows, 1, k), 0, k), View(dst.view(1, rows, 1, k), 0, k), _lib.E4M3, self.f8, i,
                                         take_amax=src is pk["wf"])

    def _fp8_ok(self, conv, x: View, cout):
        pk = self.packed.get(conv)
        return self.fp8 and pk is not None and "wf8" in pk and x.H >= 8 and x.W >= 8 and x.C % 128 == 0 and cout % 128 == 0

    def _fp8_dgrad(self, conv, dyv: View):
        """will the data gradient of `conv` (input dy) run on the fp8 kernel?"""
        pk = self.packed.get(conv)
        return (self.fp8 and pk is not None and "wd8" in pk and pk["wd"] is not None and dyv.H >= 8 and dyv.W >= 8 and dyv.C % 128 == 0 and
                pk["wd"].shape[1] % 128 == 0)

    def _q8_view(self, x: View, tag, fmt):
        key = (x.t.data_ptr(), tuple(x.t.shape))
        q = self._q8buf.get(key)
        if q is None:
            q = self._q8buf[key] = torch.empty(x.t.shape, dtype=torch.uint8, device=self.dev)
        return View(q, x.off, x.C), self.f8.slot(tag, fmt)

    def _quantised(self, x: View, tag, fmt):
        """the fp8 image of the bf16 view x and its slot: written by x's producer pass where that pass could (self._produced), else by a
        quantisation sweep here.  A slot WITHOUT history was fed its amax by the producer but scaled with a placeholder: it is calibrated
        now and the image taken again once (later steps run on the delayed scale alone)"""
        qv, i = self._q8_view(x, tag, fmt)
        if x.t.data_ptr() in self._produced and i not in self.f8.fresh:
            self._produced.discard(x.t.data_ptr())
            return qv, i
        self._produced.discard(x.t.data_ptr())
        if i in self.f8.fresh and self.f8.amax_now(i) > 0:
            h, st = _lib._load(), torch.cuda.current_stream().cuda_stream
            _lib.check(h.cvcs_fp8_update_scales(self.f8.ptr(i), 1, ops.FP8_MARGIN, st), "cvcs_fp8_update_scales")
            self.f8.fresh.discard(i)
        ops.quantize_fp8(x, qv, fmt, self.f8, i, take_amax=True)
        return qv, i

    # ------------------------------------------------------------------------------------------------ planning
    def plan(self, B, S):
        if self.shape == (B, S):
            return
        assert S % 16 == 0, "tile side must be a multiple of 16 (four 2x2 poolings)"
        dev, dt = self.dev, self.dtype
        A = lambda s, c: torch.empty((B, s, s, c), dtype=dt, device=dev)  # noqa: E731
        self.shape = (B, S)
        self._q8buf = {}
        self.sizes = [S >> l for l in range(5)]
        s_ = self.sizes
        # the 3-channel tile as NHWC rows of 16 bytes (8 bf16 / 4 f32 channels, zero padded); the first conv contracts over
        # one zero-padded K-group without fetching the padding (cvcs_conv_desc.Cin_valid)
        self.in0 = A(S, 16 // torch.empty((), dtype=dt).element_size())
        self.ya, self.aa, self.yb, self.pool, self.cat, self.x5 = {}, {}, {}, {}, {}, None
        self.g_aa, self.dy_a, self.dy_b, self.g_pool, self.g_cat = {}, {}, {}, {}, {}
        for L in range(1, 6):
            w, s = WIDTHS[L - 1], s_[L - 1]
            self.ya[L], self.aa[L], self.yb[L] = A(s, w), A(s, w), A(s, w)
            self.g_aa[L], self.dy_a[L], self.dy_b[L] = A(s, w), A(s, w), A(s, w)
            if L < 5:
                self.cat[L] = A(s, 2 * w)
                self.g_cat[L] = A(s, 2 * w)
                self.pool[L] = A(s // 2, w)
                self.g_pool[L] = A(s // 2, w)
        self.x5 = A(s_[4], WIDTHS[4])
        self.g_x5 = A(s_[4], WIDTHS[4])
        # decoder stage d works at level L = 5-d
        self.ra, self.za, self.rb, self.zb = {}, {}, {}, {}
        self.g_za, self.g_zb, self.ddy_a, self.ddy_b, self.up_in, self.g_up_in = {}, {}, {}, {}, {}, {}
        for d in range(1, 5):
            L = 5 - d
            w, s = WIDTHS[L - 1], s_[L - 1]
            self.ra[d], self.za[d], self.rb[d], self.zb[d] = A(s, w), A(s, w), A(s, w), A(s, w)
            self.g_za[d], self.g_zb[d], self.ddy_a[d], self.ddy_b[d] = A(s, w), A(s, w), A(s, w), A(s, w)
            if self.variant == "Unet":
                self.up_in[d] = A(s, 2 * w)      # bilinear-upsampled input of the upscale conv
                self.g_up_in[d] = A(s, 2 * w)
        self.logits = torch.empty((B, self.NC, S, S), dtype=torch.float32, device=dev)
        # conv-epilogue statistics scratch, sized for the worst layer
        M1 = B * S * S
        rows_c = [(ops.conv_stat_rows(View(self.ya[L], 0, WIDTHS[L - 1]), WIDTHS[L - 1], 3, 3, 1, 1), WIDTHS[L - 1])
                  for L in range(1, 6)]
        # (x2: the data gradient of a decoder's first conv has 2w channels and emits column sums for the up-conv bias)
        self.stat_sum = torch.empty(2 * max(r * c for r, c in rows_c), dtype=torch.float32, device=dev)
        self.colsum_tmp = torch.empty(2 * max(WIDTHS), dtype=torch.float32, device=dev)
        self.stat_m2 = torch.empty_like(self.stat_sum)
        self.stat_cnt = torch.empty(max(r for r, _ in rows_c), dtype=torch.float32, device=dev)
        rows1 = ops.bn_bwd_rows(M1)
        self.part = [torch.empty(rows1 * 1024, dtype=torch.float32, device=dev) for _ in range(3)]
        self.head_part = torch.empty(max(ops.head_bwd_rows(M1), ops.head_ce_rows(M1)) * (self.NC * 64 + self.NC), dtype=torch.float32, device=dev)
        # wgrad split-K workspace: the maximum over all layers
        need = 0
        for (Bc, Ho, Cout, Cin, K, st) in self._wgrad_shapes(B):
            need = max(need, ops.wgrad_workspace_floats(Bc, Ho, Ho, Cout, Cin, K, K, st))
        self.wg_ws = torch.empty(need, dtype=torch.float32, device=dev)
        bn_need = max(ops.bn_finalize_workspace_floats(r, c) for r, c in rows_c)
        self.bn_ws = torch.empty(max(bn_need, 4), dtype=torch.float32, device=dev)
        self.bn = {}
        for name in self.P:
            if name.endswith(".weight") and self.P[name].dim() == 1:
                self.bn[name[:-len(".weight")]] = _BN(self.P[name].numel(), dev)

    def _wgrad_shapes(self, B):
        s_ = self.sizes
        out = []
        for L in range(1, 6):
            w, s = WIDTHS[L - 1], s_[L - 1]
            cin = self.KG if L == 1 else WIDTHS[L - 2]
            out += [(B, s, w, cin, 3, 1), (B, s, w, w, 3, 1)]
        for d in range(1, 5):
            L = 5 - d
            w, s = WIDTHS[L - 1], s_[L - 1]
            out += [(B, s, w, 2 * w, 3, 1), (B, s, w, w, 3, 1)]
            if self.variant == "Unetv2":
                out.append((B, s // 2, 2 * w, w, 2, 2))   # ConvTranspose form: "Cout" = 2w (its Cin), "Cin" = w
            else:
                out.append((B, s, w, 2 * w, 3, 1))
        return out

    # ------------------------------------------------------------------------------------------------ helpers
    def _skip_up(self, d):
        """views of the (skip, up) channel halves of the concat buffer of decoder stage d."""
        L = 5 - d
        w = WIDTHS[L - 1]
        so, uo = (0, w) if SKIP_FIRST[d] else (w, 0)
        return View(self.cat[L], so, w), View(self.cat[L], uo, w), View(self.g_cat[L], so, w), View(self.g_cat[L], uo, w)

    @property
    def relu_order(self):
        """views of every ReLU output of the last train-mode forward, in execution order (parity tests read the path's own
        ReLU decisions from them): encoder a = relu(bn(conv)) twice per level, decoder r = relu(conv) twice per stage"""
        out = []
        for L in range(1, 6):
            out.append(ops.view(self.aa[L]))
            out.append(self._skip_up(5 - L)[0] if L < 5 else ops.view(self.x5))
        for d in range(1, 5):
            out += [ops.view(self.ra[d]), ops.view(self.rb[d])]
        return out

    def enable_sync_bn(self, sync):
        """sync: object with .world and .all_reduce(tensor) (sum over ranks, in place, ordered on the current stream)"""
        self.sync_bn = sync
        self._sync_mom = torch.empty(3 * 1024, dtype=torch.float64, device=self.dev)
        self._sync_sums = torch.empty(2 * 1024, dtype=torch.float32, device=self.dev)

    def _conv_bn(self, x: View, conv, bnname, y, out: View, relu_after_bn: bool, train: bool, pool: View | None = None,
                 apply: bool = True, q8_for: str | None = None):
        """conv3x3(+bias) [-> ReLU] with fused statistics, BN finalize, BN apply [-> ReLU] [-> pool]."""
        pk = self.packed[conv]
        ops.SCOPE = "enc" if conv.startswith("encode") else "dec"
        M = x.B * y.shape[1] * y.shape[2]
        C_ = y.shape[3]
        st = self.bn[bnname]
        if not train:
            # eval mode: the BatchNorm is a fixed per-channel affine -> folded into the conv epilogue (before the ReLU in the
            # encoder order conv->BN->ReLU, after it in the decoder order conv->ReLU->BN); the activation is written once,
            # and only pooled layers need a second (pooling-only) pass
            ops.bn_finalize(None, 0, M, C_, self.P[bnname + ".weight"], self.P[bnname + ".bias"],
                            self.Bf[bnname + ".running_mean"], self.Bf[bnname + ".running_var"], False, st.scale, st.shift,
                            None, None)
            fold = (st.scale, st.shift)
            fuse_pool = pool is not None and x.H >= 8 and x.W >= 8      # the halo kernel pools in its epilogue
            ops.conv2d(x, pk["wf"], self.P[conv + ".bias"], out, 3, 3, 1, 1, relu=True,
                       pre_affine=fold if relu_after_bn else None, post_affine=None if relu_after_bn else fold,
                       cin_real=3 if conv == "encode1.0.layer.0" else None, pool=pool if fuse_pool else None)
            if pool is not None and not fuse_pool:
                ops.bn_act(out, self.one[:C_], self.zero[:C_], False, None, pool)
            return
        stats = (self.stat_sum, self.stat_m2, self.stat_cnt)
        if self._fp8_ok(conv, x, C_):
            rows = ops.fp8_stat_rows(x)
            x8, ix = self._quantised(x, "a:" + conv, _lib.E4M3)
            # (the conv bias rides in the epilogue's per-channel affine: v = acc * s_x * s_w * 1 + bias)
            ops.conv3x3_fp8(x8, _lib.E4M3, pk["wf8"], ops.view(y), self.f8, ix, self.f8.slot("w:" + conv, _lib.E4M3), stats=stats,
                            relu=not relu_after_bn, pre_affine=(self.one[:C_], self.P[conv + ".bias"]))
        else:
            rows = ops.conv_stat_rows(x, C_, 3, 3, 1, 1)
            ops.conv2d(x, pk["wf"], self.P[conv + ".bias"], ops.view(y), 3, 3, 1, 1, relu=not relu_after_bn, stats=stats,
                       cin_real=3 if conv == "encode1.0.layer.0" else None)
        if self.sync_bn is None:
            ops.bn_finalize(stats, rows, M, C_, self.P[bnname + ".weight"], self.P[bnname + ".bias"],
                            self.Bf[bnname + ".running_mean"], self.Bf[bnname + ".running_var"], True, st.scale, st.shift,
                            st.mean, st.invstd, workspace=self.bn_ws)
        else:
            self._sync_bn_finalize(stats, rows, C_, bnname, st)
        if apply:   # apply=False: the consumer folds scale/shift into its weights (the 1x1 head)
            # fp8: an activation that the next conv reads as e4m3 leaves this pass with its fp8 image (no quantisation sweep of its own);
            # pooled layers (the pass writes two tensors) keep the separate sweep
            q8 = None
            if self.fp8 and pool is None and q8_for is not None and out.C % 128 == 0 and out.H >= 8 and out.W >= 8 and out.off == 0 and out.C == out.ld:
                qv, i = self._q8_view(out, "a:" + q8_for, _lib.E4M3)
                q8 = (qv, _lib.E4M3, self.f8, i, True)
                self._produced.add(out.t.data_ptr())
            ops.bn_act(ops.view(y), st.scale, st.shift, relu_after_bn, out, pool, q8=q8)

    def _sync_bn_finalize(self, stats, rows, C_, bnname, st):
        # batch statistics over the tiles of ALL ranks (the reference's batch is not sharded): one [3][C] f64 sum
        mom = self._sync_mom[:3 * C_]
        ops.bn_moments(stats, rows, C_, mom, workspace=self.bn_ws)
        self.sync_bn.all_reduce(mom)
        ops.bn_finalize_moments(mom, C_, self.P[bnname + ".weight"], self.P[bnname + ".bias"],
                                self.Bf[bnname + ".running_mean"], self.Bf[bnname + ".running_var"], st.scale, st.shift,
                                st.mean, st.invstd)

    # ------------------------------------------------------------------------------------------------ forward
    def _backbone(self, x: torch.Tensor, train: bool) -> View:
        """everything in front of the 1x1 head: x NCHW u8 | f32 [B,3,S,S] -> the last decoder activation (NHWC view)"""
        B, C_, S, S2 = x.shape
        assert C_ == 3 and S == S2
        self.plan(B, S)
        self._produced.clear()
        self.refresh_weights(train)
        ops.pack_input(x.contiguous(), self.in0)
        cur = ops.view(self.in0)
        for L in range(1, 6):
            base = 0 if L == 1 else 1
            ca, cb = f"encode{L}.{base}.layer.0", f"encode{L}.{base + 1}.layer.0"
            self._conv_bn(cur, ca, f"encode{L}.{base}.layer.1", self.ya[L], ops.view(self.aa[L]), True, train, q8_for=cb)
            if L < 5:
                d = 5 - L
                skip_v = self._skip_up(d)[0]
                self._conv_bn(ops.view(self.aa[L]), cb, f"encode{L}.{base + 1}.layer.1", self.yb[L], skip_v, True, train,
                              pool=ops.view(self.pool[L]))
                cur = ops.view(self.pool[L])
            else:
                self._conv_bn(ops.view(self.aa[L]), cb, f"encode{L}.{base + 1}.layer.1", self.yb[L], ops.view(self.x5), True, train)
        prev = ops.view(self.x5)
        ops.SCOPE = "dec"
        for d in range(1, 5):
            L = 5 - d
            _, up_v, _, _ = self._skip_up(d)
            if self.variant == "Unetv2":
                pk = self.packed[f"upscale{d}.0"]
                ops.conv2d(prev, pk["wf"], pk["b4"], up_v, 1, 1, pixel_shuffle=True)
            else:
                ops.upsample2x_fwd(prev, ops.view(self.up_in[d]))
                key = f"upscale{d}.0.layer.1"
                ops.conv2d(ops.view(self.up_in[d]), self.packed[key]["wf"], self.P[key + ".bias"], up_v, 3, 3, 1, 1)
            pre = f"decode_forward{d}.0.layer"
            self._conv_bn(ops.view(self.cat[L]), pre + ".0", pre + ".2", self.ra[d], ops.view(self.za[d]), False, train, q8_for=pre + ".3")
            # train mode: the very last BatchNorm is followed only by the 1x1 head -> its apply pass (over the largest
            # activation) is skipped, forward() folds the affine into the head weights
            fold_head = train and d == 4
            self._conv_bn(ops.view(self.za[d]), pre + ".3", pre + ".5", self.rb[d], ops.view(self.zb[d]), False, train,
                          apply=not fold_head)
            prev = ops.view(self.rb[d]) if fold_head else ops.view(self.zb[d])
        self._saved_train = train
        return prev

    def forward(self, x: torch.Tensor, train: bool, lazy: bool = False) -> torch.Tensor:
        """x: NCHW u8 | f32 [B,3,S,S] on the device -> NCHW f32 logits [B,NC,S,S] (owned by the engine).
        lazy (train mode, bf16): the head is NOT run - materialize_logits() fills the buffer if anybody reads it, a loss that recognises the
        handle calls head_ce() and the logits never exist (nets.LazyLogits; S/train.py:121-125 unchanged)"""
        prev = self._backbone(x, train)
        hw = self.P["decode_forward4.1.weight"].view(self.NC, 64)
        hb = self.P["decode_forward4.1.bias"]
        if train:
            st = self.bn["decode_forward4.0.layer.5"]
            ops.head_fold(hw, hb, st.scale, st.shift, self.head_wf, self.head_bf)
            hw, hb = self.head_wf, self.head_bf
        self.fwd_serial += 1
        self._head_done = False
        self._head_in, self._head_w = prev, (hw, hb)
        self._head_pending = bool(lazy and train and self.lazy_head_ok())
        if not self._head_pending:
            ops.head_fwd(prev, hw, hb, self.logits)
        return self.logits

    def lazy_head_ok(self) -> bool:
        return self.dtype == torch.bfloat16 and self.NC <= 32 and self.sync_bn is None

    def materialize_logits(self):
        if self._head_pending:
            ops.head_fwd(self._head_in, self._head_w[0], self._head_w[1], self.logits)
            self._head_pending = False

    def head_ce(self, target, class_weight, ignore_index, workspace, loss_out):
        """folded head + cross-entropy + the head's backward in one pass over the last ReLU output (cvcs_head_ce): logits from the folded
        weights, the gradient w.r.t. the BatchNorm output from the real ones, dW | db partials w.r.t. the folded ones (un-folded in backward)"""
        B, S = self.shape
        P_ = B * S * S
        rows = ops.head_ce_rows(P_)
        assert self.head_part.numel() >= rows * (self.NC * 64 + self.NC)
        ops.head_ce(self._head_in, self._head_w[0], self._head_w[1], target, class_weight, ignore_index, 1.0, ops.view(self.g_zb[4]),
                    self.head_part, workspace, loss_out, w_dx=self.P["decode_forward4.1.weight"].view(self.NC, 64))
        self._head_done, self._head_rows = True, rows

    def scale_head_grads(self, gout):
        g = gout.detach().to(device=self.head_part.device, dtype=torch.float32).reshape(1).contiguous()
        ops.scale_unless_one(self.g_zb[4], g)
        ops.scale_unless_one(self.head_part, g)

    def forward_labels(self, x: torch.Tensor, labels: torch.Tensor):
        """eval-mode prediction: u8 labels [B,S,S] = argmax of the logits, head and argmax fused (no logits tensor)"""
        prev = self._backbone(x, False)
        hw = self.P["decode_forward4.1.weight"]
        ops.head_argmax(prev, hw.view(self.NC, 64), self.P["decode_forward4.1.bias"], labels)
        return labels

    # ------------------------------------------------------------------------------------------------ backward
    def _fused_reduce(self, bnname, y, mode, dgrad_conv=None, dy=None):
        """argument of ops.conv2d(bn_bwd=...) for the data-gradient launch whose output is the gradient this BatchNorm
        consumes - or None where the reduce pass stays its own launch (f32, maps under 8 pixels, CVCS_FUSE_BN_BWD=0)"""
        yv = ops.view(y)
        if not self.fuse_bn_bwd or self.dtype != torch.bfloat16 or yv.H < 8 or yv.W < 8:
            return None
        if self.fp8 and dgrad_conv is not None and self._fp8_dgrad(dgrad_conv, ops.view(dy)):
            return None        # (the fp8 kernel has no fused reduce epilogue)
        if not (self.fuse_bn_bwd_c[0] <= yv.C <= self.fuse_bn_bwd_c[1]):
            return None
        st = self.bn[bnname]
        return (yv, st.scale, st.shift, st.mean, st.invstd, mode, self.stat_sum, self.stat_m2)

    def _bn_backward(self, bnname, conv, y, g1: View, g2, mode, dy, fused=False):
        """two-pass BN(+ReLU)(+pool) backward; fills dgamma, dbeta, conv-bias gradient; writes dy.
        fused: the first pass was taken by the launch that produced g1 (one partial row per 16x16 tile in
        stat_sum / stat_m2); only the row reduction is left of it."""
        st = self.bn[bnname]
        yv = ops.view(y)
        M = yv.B * yv.H * yv.W
        C_ = yv.C
        rows = ops.bn_bwd_rows(M)
        p0, p1, p2 = (p[:rows * C_] for p in self.part)
        if fused:
            trows = yv.B * ((yv.H + 15) // 16) * ((yv.W + 15) // 16)
            rows1 = ops.bn_bwd_rows(trows)
            for src, dst in ((self.stat_sum, p0), (self.stat_m2, p1)):
                ops.colsum_partial(View(src[:trows * C_].view(1, trows, 1, C_), 0, C_), dst[:rows1 * C_])
            ops.bn_bwd_finalize(p0, p1, rows1, M, C_, self.P[bnname + ".weight"], st.invstd, self.G[bnname + ".weight"],
                                self.G[bnname + ".bias"], st.ca, st.cb)
        else:
            ops.bn_bwd_reduce(yv, g1, g2, st.scale, st.shift, st.mean, st.invstd, mode, p0, p1)
            ops.bn_bwd_finalize(p0, p1, rows, M, C_, self.P[bnname + ".weight"], st.invstd, self.G[bnname + ".weight"],
                                self.G[bnname + ".bias"], st.ca, st.cb)
        if self.sync_bn is not None:
            # dgamma / dbeta stay this rank's sums (the gradient all-reduce adds them up); the two coefficients of
            # pass 2 are means over every rank's pixels
            sums = self._sync_sums[:2 * C_]
            sums[:C_].copy_(self.G[bnname + ".bias"])
            sums[C_:].copy_(self.G[bnname + ".weight"])
            self.sync_bn.all_reduce(sums)
            ops.bn_bwd_coeffs(sums, M * self.sync_bn.world, C_, st.ca, st.cb)
        q8 = None
        dyv = ops.view(dy)
        if self.fp8 and g2 is None and self._fp8_dgrad(conv, dyv):
            qv, i = self._q8_view(dyv, "g:" + conv, _lib.E5M2)
            q8 = (qv, _lib.E5M2, self.f8, i, True)
            self._produced.add(dy.data_ptr())
        ops.bn_bwd_apply(yv, g1, g2, st.scale, st.shift, st.mean, st.invstd, st.ca, st.cb, mode, dyv, p2, q8=q8)
        ops.colsum_finalize(p2, rows, C_, self.G[conv + ".bias"])

    def _wgrad(self, x: View, dy: View, dw, KH, KW, stride, pad, cin_real=None):
        """weight gradient on the side stream (ordered after everything issued so far on the main stream; the side
        stream itself serialises the launches that share the split-K workspace)"""
        if not self.overlap_wgrad:
            ops.conv2d_wgrad(x, dy, dw, KH, KW, stride, pad, self.wg_ws, cin_real=cin_real)
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._side.wait_event(ev)
        with torch.cuda.stream(self._side):
            ops.conv2d_wgrad(x, dy, dw, KH, KW, stride, pad, self.wg_ws, cin_real=cin_real)
            self._side_event = torch.cuda.Event()
            self._side_event.record(self._side)

    def _conv_backward(self, conv, x: View, dy, gin: View | None, cin_real=None, colsum_of: View | None = None, colsum_out=None,
                       bn_bwd=None):
        """weight and data gradient of a 3x3 conv.  colsum_of: a channel range of `gin` (the up-sampled half of a concat
        gradient) whose per-channel sum over pixels - the bias gradient of the up-conv that produced it - is wanted: the
        dgrad launch emits the column sums of its output through the statistics epilogue, no extra pass over the tensor."""
        dyv = ops.view(dy)
        ops.SCOPE = "enc" if conv.startswith("encode") else "dec"
        self._wgrad(x, dyv, self.G[conv + ".weight"], 3, 3, 1, 1, cin_real=cin_real)
        if gin is None:
            return
        f8 = self._fp8_ok(conv, dyv, gin.C)
        if f8:    # e5m2 gradient (range over precision) x e4m3 flipped weights
            dy8, ig = self._quantised(dyv, "g:" + conv, _lib.E5M2)
            iw = self.f8.slot("w:" + conv, _lib.E4M3)
        if colsum_of is None:
            if f8:
                assert bn_bwd is None
                ops.conv3x3_fp8(dy8, _lib.E5M2, self.packed[conv]["wd8"], gin, self.f8, ig, iw)
            else:
                ops.conv2d(dyv, self.packed[conv]["wd"], None, gin, 3, 3, 1, 1, bn_bwd=bn_bwd)
            return
        assert bn_bwd is None
        C2 = gin.C
        if f8:
            rows = ops.fp8_stat_rows(dyv)
            ops.conv3x3_fp8(dy8, _lib.E5M2, self.packed[conv]["wd8"], gin, self.f8, ig, iw, stats=(self.stat_sum, self.stat_m2, self.stat_cnt))
        else:
            rows = ops.conv_stat_rows(dyv, C2, 3, 3, 1, 1)
            ops.conv2d(dyv, self.packed[conv]["wd"], None, gin, 3, 3, 1, 1, stats=(self.stat_sum, self.stat_m2, self.stat_cnt))
        if rows > 2048:   # many partial rows: a parallel first stage (the one-launch finalize has only C/16 workgroups)
            prow = ops.bn_bwd_rows(rows)
            ops.colsum_partial(View(self.stat_sum[:rows * C2].view(1, rows, 1, C2), 0, C2), self.part[0][:prow * C2])
            ops.colsum_finalize(self.part[0], prow, C2, self.colsum_tmp[:C2])
        else:
            ops.colsum_finalize(self.stat_sum, rows, C2, self.colsum_tmp[:C2])
        lo = colsum_of.off - gin.off
        colsum_out.copy_(self.colsum_tmp[lo:lo + colsum_of.C])

    def backward(self, dlogits: torch.Tensor):
        """dlogits: NCHW f32 [B,NC,S,S]; fills every gradient view in self.G (overwrites)."""
        assert self._saved_train, "backward needs a preceding forward in train mode"
        if self.on_backward_begin is not None:
            self.on_backward_begin()
        def ready(name):
            if self.on_grad_ready is not None:   # the bucket must also wait for the side-stream weight gradients
                self.on_grad_ready(self._goff[name], [self._side_event] if self._side_event is not None else [])
        B, S = self.shape
        NC = self.NC
        hw = self.P["decode_forward4.1.weight"]
        if self._head_done:       # head_ce() already produced the gradient w.r.t. the BatchNorm output and the partial rows
            rows = self._head_rows
            self._head_done = False
        else:
            rows = ops.head_bwd_rows(B * S * S)
            # the head saw r (the last ReLU output) through folded weights: dx = W^T dl is the gradient w.r.t. the BatchNorm
            # output z either way (real W); dW comes out w.r.t. r and is un-folded below
            ops.head_bwd(ops.view(self.rb[4]), dlogits.contiguous(), hw.view(NC, 64), ops.view(self.g_zb[4]), self.head_part)
        # head weight [NC,64,1,1] and bias [NC] are adjacent in the flat gradient buffer (registration order)
        gw, gb = self.G["decode_forward4.1.weight"], self.G["decode_forward4.1.bias"]
        assert gb.data_ptr() == gw.data_ptr() + gw.numel() * 4
        ops.colsum_finalize(self.head_part, rows, NC * 64 + NC, gw)
        st = self.bn["decode_forward4.0.layer.5"]
        ops.head_unfold_grad(gw, gb, st.scale, st.shift)
        ready("decode_forward4.1.weight")
        g = ops.view(self.g_zb[4])
        for d in range(4, 0, -1):
            L = 5 - d
            w = WIDTHS[L - 1]
            pre = f"decode_forward{d}.0.layer"
            _, up_v, gskip_v, gup_v = self._skip_up(d)
            self._bn_backward(pre + ".5", pre + ".3", self.rb[d], g, None, 1, self.ddy_b[d])
            fz = self._fused_reduce(pre + ".2", self.ra[d], 1, pre + ".3", self.ddy_b[d])
            self._conv_backward(pre + ".3", ops.view(self.za[d]), self.ddy_b[d], ops.view(self.g_za[d]), bn_bwd=fz)
            self._bn_backward(pre + ".2", pre + ".0", self.ra[d], ops.view(self.g_za[d]), None, 1, self.ddy_a[d], fused=fz is not None)
            key = f"upscale{d}.0" if self.variant == "Unetv2" else f"upscale{d}.0.layer.1"
            self._conv_backward(pre + ".0", ops.view(self.cat[L]), self.ddy_a[d], ops.view(self.g_cat[L]),
                                colsum_of=gup_v, colsum_out=self.G[key + ".bias"])
            ready(pre + ".0.weight")
            prev = ops.view(self.x5) if d == 1 else ops.view(self.zb[d - 1])
            gprev = ops.view(self.g_x5) if d == 1 else ops.view(self.g_zb[d - 1])
            if self.variant == "Unetv2":
                # dW[ci][co][2][2] = sum_p in[p][ci] * gup[2p+tap][co]: the "output gradient" role is played by the input
                self._wgrad(gup_v, prev, self.G[key + ".weight"], 2, 2, 2, 0)
                ops.conv2d(gup_v, self.packed[key]["wd"], None, gprev, 2, 2, stride=2, pad=0)
            else:
                self._wgrad(ops.view(self.up_in[d]), gup_v, self.G[key + ".weight"], 3, 3, 1, 1)
                ops.conv2d(gup_v, self.packed[key]["wd"], None, ops.view(self.g_up_in[d]), 3, 3, 1, 1)
                ops.upsample2x_bwd(ops.view(self.g_up_in[d]), gprev)
            ready(key + ".weight")
            g = gprev
        # encoder, deepest level first; g = d(x5)
        for L in range(5, 0, -1):
            base = 0 if L == 1 else 1
            ca, cb = f"encode{L}.{base}.layer.0", f"encode{L}.{base + 1}.layer.0"
            if L == 5:
                g1, g2 = g, None
            else:
                g1, g2 = self._skip_up(5 - L)[2], ops.view(self.g_pool[L])
            self._bn_backward(f"encode{L}.{base + 1}.layer.1", cb, self.yb[L], g1, g2, 0, self.dy_b[L])
            fz = self._fused_reduce(f"encode{L}.{base}.layer.1", self.ya[L], 0, cb, self.dy_b[L])
            self._conv_backward(cb, ops.view(self.aa[L]), self.dy_b[L], ops.view(self.g_aa[L]), bn_bwd=fz)
            ready(cb + ".weight")
            self._bn_backward(f"encode{L}.{base}.layer.1", ca, self.ya[L], ops.view(self.g_aa[L]), None, 0, self.dy_a[L],
                              fused=fz is not None)
            if L == 1:
                self._conv_backward(ca, ops.view(self.in0), self.dy_a[L], None, cin_real=3)
            else:
                self._conv_backward(ca, ops.view(self.pool[L - 1]), self.dy_a[L], ops.view(self.g_pool[L - 1]))
            ready(ca + ".weight")
        if self._side is not None:   # the optimiser and the next forward run after every weight gradient has landed
            torch.cuda.current_stream().wait_stream(self._side)
        if self.fp8:
            self.f8.update()      # delayed scaling: this step's amax of every quantised tensor becomes the next step's scale
