"""Launch plan of Swin-T + UPerNet (BASELINE.json configs[3]: "Swin-T encoder + UPerNet decoder, 512x512 window-7 attention";
north star: "patch-embed + windowed attention for the ViT/Swin encoder variant") on the HIP kernels.

The reference documents `TSwin` / `BSwin` (configs/train/README.txt:44-45) but ships no definition; the architecture is the
published one, with the structure and parameter names of transformers' UperNetForSemanticSegmentation over a SwinBackbone (the CPU
checker states it in plain torch and is cross-checked against that implementation).

MI355X-first choices:
  * tokens ARE NHWC pixels: a token row = the C channels of one pixel, so every nn.Linear is a 1x1 convolution on the GEMM kernels
    (taps kernel forward / data gradient with the bias in the accumulators, `wgrad_gemm_kernel` weight gradient), no reshapes;
  * q / k / v projections are ONE GEMM: their three weights (and biases) are adjacent in the flat parameter buffer, so the fused
    [3C, C] operand, its gradient and the optimiser see plain contiguous memory;
  * window partition, zero padding to a multiple of 7 and the cyclic shift are index arithmetic in one gather; the inverse is fused
    with the residual add; the -100 region mask and the relative-position bias are computed inside the attention kernel;
  * the backward of attention recomputes the probabilities (nothing of size tokens x 49 is stored);
  * UPerNet: conv + BN(+statistics in the conv epilogue) + ReLU units of the ResNet plan; PPM outputs, the last stage and the four
    FPN outputs are written straight into channel ranges of the two concat buffers;
  * recorded launch plans as in resnet_engine.py.
"""
from __future__ import annotations

import contextlib

import torch

from . import _lib, ops
from .ops import View
from .resnet_engine import Act, ResNetUNetEngine

EMBED, DEPTHS, HEADS, WINDOW, HIDDEN, POOL_SCALES = 96, (2, 2, 6, 2), (3, 6, 12, 24), 7, 512, (1, 2, 3, 6)
DIMS = tuple(EMBED * 2 ** i for i in range(4))
# (embed dim, depths, heads): Swin-T (the module constants above) and Swin-B (`BSwin`, configs/train/README.txt:45); head dim 32 in both
VARIANTS = {"tiny": (EMBED, DEPTHS, HEADS), "base": (128, (2, 2, 18, 2), (4, 8, 16, 32))}
PE = "backbone.swin.embeddings.patch_embeddings.projection"


class SwinUPerNetEngine(ResNetUNetEngine):
    def __init__(self, num_classes: int, dtype: torch.dtype, device, variant="tiny", fp8=False):
        # fp8: the UPerNet head's 3x3 convolutions (bottleneck 2816 -> 512, three FPN 512 -> 512, the 2048 -> 512 fusion conv: every channel
        # count a multiple of 128) run forward and data gradient on the fp8 kernel (resnet_engine._unit / _unit_bwd / _dgrad); everything
        # else - the transformer, the 1x1 laterals, all weight gradients - stays bf16
        super().__init__("resnet18", num_classes, dtype, device, fp8=fp8)   # (arch only sizes helper vectors; no ResNet stage is built)
        self.head_name = "decode_head.classifier"
        self.embed, self.swin_depths, self.swin_heads = VARIANTS[variant]
        self.dims = tuple(self.embed * 2 ** i for i in range(4))

    # ------------------------------------------------------------------------------------------------ binding
    def bind(self, params, grads, buffers):
        self.P, self.G, self.Bf = params, grads, buffers
        base = min(g.data_ptr() for g in grads.values())
        self._goff = {k: (g.data_ptr() - base) // 4 for k, g in grads.items()}
        dev, dt = self.dev, self.dtype
        self.packed, entries = {}, []
        self.W4, self.GW4 = {}, {}          # 4-D (OIHW) views of the linear / conv weights and of their gradients

        def add(key, w4, g4, want_dgrad=True, cin_pad=None):
            cout, cin, kh, kw = w4.shape
            cp = cin_pad or cin
            pk = dict(wf=torch.empty(kh * kw, cout, cp, dtype=dt, device=dev),
                      wd=torch.empty(kh * kw, cin, cout, dtype=dt, device=dev) if want_dgrad else None)
            self.packed[key], self.W4[key], self.GW4[key] = pk, w4, g4
            entries.append((w4, pk["wf"], pk["wd"]))

        for name, w in params.items():
            if not name.endswith(".weight") or name == self.head_name + ".weight":
                continue
            key = name[:-len(".weight")]
            if key.endswith(".q_proj"):
                # fused q | k | v: the three [C, C] weights are adjacent in the flat buffers (nets.swin_upernet_param_spec)
                C_ = w.shape[0]
                att = key[:-len(".q_proj")]
                for n_, off in (("k_proj", 1), ("v_proj", 2)):
                    assert params[f"{att}.{n_}.weight"].data_ptr() == w.data_ptr() + off * C_ * C_ * 4
                    assert params[f"{att}.{n_}.bias"].data_ptr() == params[f"{att}.q_proj.bias"].data_ptr() + off * C_ * 4
                fuse = lambda t, shape: torch.as_strided(t, shape, tuple(int(torch.tensor(shape[i + 1:]).prod()) for i in range(len(shape))))  # noqa: E731
                add(att + ".qkv", fuse(w, (3 * C_, C_, 1, 1)), fuse(grads[name], (3 * C_, C_, 1, 1)))
                self.W4[att + ".qkv.bias"] = fuse(params[f"{att}.q_proj.bias"], (3 * C_,))
                self.GW4[att + ".qkv.bias"] = fuse(grads[f"{att}.q_proj.bias"], (3 * C_,))
            elif key.endswith((".k_proj", ".v_proj")):
                continue
            elif key == PE:
                add(key, w.view(self.embed, 48, 1, 1), grads[name].view(self.embed, 48, 1, 1), want_dgrad=False, cin_pad=64)
            elif w.dim() == 2:
                add(key, w.view(w.shape[0], w.shape[1], 1, 1), grads[name].view(w.shape[0], w.shape[1], 1, 1))
            elif w.dim() == 4:
                add(key, w, grads[name])
        self._pack_table = ops.pack_table(entries, dev)
        if self.fp8:
            self.f8 = ops.Fp8Slots(dev)
            for name, pk in self.packed.items():
                t_, cout, cin = pk["wf"].shape
                if t_ == 9 and cout % 128 == 0 and cin % 128 == 0 and pk["wd"] is not None:
                    pk["wf8"], pk["wd8"] = (torch.empty(t.shape, dtype=torch.uint8, device=dev) for t in (pk["wf"], pk["wd"]))
        from .engine import _BN
        self.bn = {n[:-len(".weight")]: _BN(p.numel(), dev) for n, p in params.items() if n.endswith("batch_norm.weight")}
        self.one = torch.ones(4096, dtype=torch.float32, device=dev)
        self.zero = torch.zeros(4096, dtype=torch.float32, device=dev)
        self._rec, self._bufs, self.shape, self.gn = {}, {}, None, {}
        self._shape_cache.clear()

    def refresh_weights(self, train=True):
        ops.pack_conv_weights(self._pack_table[0], self._pack_table[1], self.dtype)
        if self.fp8:
            for name, pk in self.packed.items():
                if "wf8" in pk:
                    i = self.f8.slot("w:" + name, _lib.E4M3)
                    for src, dst in ((pk["wf"], pk["wf8"]), (pk["wd"], pk["wd8"])):
                        rows, k = src.shape[0] * src.shape[1], src.shape[2]
                        ops.quantize_fp8(View(src.view(1, rows, 1, k), 0, k), View(dst.view(1, rows, 1, k), 0, k), _lib.E4M3, self.f8, i,
                                         take_amax=train and src is pk["wf"])

    # ------------------------------------------------------------------------------------------------ small pieces
    def _tokens(self, name, B, H, W, C_):
        return ops.view(self._act(name, B, H, W, C_))

    def _add(self, a: View, b: View, out: View):
        ops.relu_bwd_sum(None, [(a, False), (b, False)], out)

    def _lin(self, x: View, key, out: View, bias=True):
        """nn.Linear as a 1x1 convolution over the token map x -> out (bias inside the accumulators)"""
        ops.SCOPE = "enc" if key.startswith("backbone") else "dec"
        b = self.W4.get(key + ".bias") if bias else None
        if bias and b is None:
            b = self.P[key + ".bias"]
        ops.conv2d(x, self.packed[key]["wf"], b, out, 1, 1, cin_real=48 if key == PE else None)

    def _lin_bwd(self, x: View, g: View, key, gx_name, bias=True, want_dx=True):
        """weight (and bias) gradient of a linear layer; returns the data gradient"""
        ops.SCOPE = "enc" if key.startswith("backbone") else "dec"
        gw = self.GW4[key]
        gb = None
        if bias:
            gb = self.GW4.get(key + ".bias")
            if gb is None:
                gb = self.G[key + ".bias"]
        # the bias gradient (column sums of g) comes out of the weight-gradient GEMM, which streams g anyway (bf16; the f32 path sums apart)
        fused_bias = bias and ops.wgrad_takes_bias(x, g, 1, 1, 1, 0)
        need = ops.wgrad_workspace_floats_for(x, g, 1, 1, 1, 0, with_bias=fused_bias)
        # (x and g are per-layer buffers that nothing overwrites during this backward: no hazard on the side lane)
        side = self.overlap_wgrad and _lib._recording is not None
        with (_lib.side_lane() if side else contextlib.nullcontext()):
            ops.conv2d_wgrad(x, g, gw, 1, 1, 1, 0, self._scratch("wg_ws_side" if side else "wg_ws", need), cin_real=48 if key == PE else None,
                             dbias=gb if fused_bias else None)
        if bias and not fused_bias:
            M = g.B * g.H * g.W
            rows = ops.bn_bwd_rows(M)
            part = self._scratch("bias_part", rows * g.C)
            ops.colsum_partial(g, part)
            ops.colsum_finalize(part, rows, g.C, gb)
        if not want_dx:
            return None
        gx = self._tokens(gx_name, x.B, x.H, x.W, x.C)
        ops.conv2d(g, self.packed[key]["wd"], None, gx, 1, 1)
        return gx

    def _ln(self, x: View, name, out: View):
        M = x.B * x.H * x.W
        st = self.gn.setdefault(name + f"@{M}", dict(mean=torch.empty(M, dtype=torch.float32, device=self.dev),
                                                      invstd=torch.empty(M, dtype=torch.float32, device=self.dev)))
        ops.layernorm_fwd(x, self.P[name + ".weight"], self.P[name + ".bias"], out, st["mean"], st["invstd"])
        return st

    def _ln_bwd(self, x: View, g: View, name, st, dx: View):
        M = x.B * x.H * x.W
        rows = ops.layernorm_rows(M)
        part = self._scratch("ln_part", rows * 2 * x.C)
        ops.layernorm_bwd(x, g, self.P[name + ".weight"], st["mean"], st["invstd"], dx, part)
        gw, gb = self.G[name + ".weight"], self.G[name + ".bias"]
        assert gb.data_ptr() == gw.data_ptr() + x.C * 4       # (gamma | beta) adjacent: one finalize over 2C columns
        ops.colsum_finalize(part, rows, 2 * x.C, gw)

    # ------------------------------------------------------------------------------------------------ Swin block
    def _block(self, x: Act, p, heads, shift, out: View, train, tape) -> Act:   # noqa: D401  (overrides the ResNet block)
        B, H, W, C_ = x.v.B, x.v.H, x.v.W, x.v.C
        Hp, Wp = -(-H // WINDOW) * WINDOW, -(-W // WINDOW) * WINDOW
        T = B * Hp * Wp
        att = p + ".attention"
        table = self.P[att + ".relative_position_bias.relative_position_bias_table"]
        h1 = self._tokens(p + ".h1", B, H, W, C_)
        st1 = self._ln(x.v, p + ".layernorm_before", h1)
        win = self._tokens(p + ".win", 1, T, 1, C_)
        ops.window_gather(h1, win, shift)
        qkv = self._tokens(p + ".qkv", 1, T, 1, 3 * C_)
        self._lin(win, att + ".qkv", qkv)
        ao = self._tokens(p + ".ao", 1, T, 1, C_)
        ops.window_attention_fwd(qkv, B, H, W, heads, shift, table, ao)
        pr = self._tokens(p + ".proj", 1, T, 1, C_)
        self._lin(ao, att + ".o_proj", pr)
        x1 = self._tokens(p + ".x1", B, H, W, C_)
        ops.window_reverse(pr, x.v, x1, shift)
        h2 = self._tokens(p + ".h2", B, H, W, C_)
        st2 = self._ln(x1, p + ".layernorm_after", h2)
        f1 = self._tokens(p + ".f1", B, H, W, 4 * C_)
        self._lin(h2, p + ".mlp.fc1", f1)
        a = self._tokens(p + ".gelu", B, H, W, 4 * C_)
        ops.gelu(f1, a)
        f2 = self._tokens(p + ".f2", B, H, W, C_)
        self._lin(a, p + ".mlp.fc2", f2)
        self._add(x1, f2, out)
        o = Act(out)
        if train:
            def bwd():
                g = self._sum_grads(o, p + ".g_out")
                g_a = self._lin_bwd(a, g, p + ".mlp.fc2", p + ".g_a")
                g_f1 = self._tokens(p + ".g_f1", B, H, W, 4 * C_)
                ops.gelu(f1, g_f1, g=g_a)
                g_h2 = self._lin_bwd(h2, g_f1, p + ".mlp.fc1", p + ".g_h2")
                g_x1b = self._tokens(p + ".g_x1b", B, H, W, C_)
                self._ln_bwd(x1, g_h2, p + ".layernorm_after", st2, g_x1b)
                g_x1 = self._tokens(p + ".g_x1", B, H, W, C_)
                self._add(g, g_x1b, g_x1)
                g_pr = self._tokens(p + ".g_proj", 1, T, 1, C_)
                ops.window_gather(g_x1, g_pr, shift)                       # transpose of the reverse (padding tokens: 0)
                g_ao = self._lin_bwd(ao, g_pr, att + ".o_proj", p + ".g_ao")
                g_qkv = self._tokens(p + ".g_qkv", 1, T, 1, 3 * C_)
                ws = self._scratch("attn_ws", ops.window_attention_bwd_workspace(B, H, W, heads))
                ops.window_attention_bwd(qkv, g_ao, B, H, W, heads, shift, table, g_qkv,
                                         self.G[att + ".relative_position_bias.relative_position_bias_table"], ws)
                g_win = self._lin_bwd(win, g_qkv, att + ".qkv", p + ".g_win")
                g_h1 = self._tokens(p + ".g_h1", B, H, W, C_)
                ops.window_reverse(g_win, None, g_h1, shift)               # transpose of the gather
                g_xa = self._tokens(p + ".g_xa", B, H, W, C_)
                self._ln_bwd(x.v, g_h1, p + ".layernorm_before", st1, g_xa)
                g_x = self._tokens(p + ".g_x", B, H, W, C_)
                self._add(g_x1, g_xa, g_x)
                x.grads.append((g_x, False))
                self._ready(att + ".q_proj.weight")
            tape.append(bwd)
        return o

    def _sum_grads(self, a: Act, name) -> View:
        """the gradient of an activation with one to three consumers as ONE view"""
        assert 1 <= len(a.grads) <= 3 and not any(h for _, h in a.grads)
        if len(a.grads) == 1:
            return a.grads[0][0]
        out = self._tokens(name, a.v.B, a.v.H, a.v.W, a.v.C)
        ops.relu_bwd_sum(None, a.grads, out)
        return out

    # ------------------------------------------------------------------------------------------------ the network
    def _backbone(self, B, S, train):
        tape = []
        self.units, self.relu_order = {}, []
        self._q8, self._train_pass = {}, train
        self.refresh_weights(train)
        H = S // 4
        h4, h1 = S // 32, S // 4
        # ---- patch embedding (the packed patches come from the eager cvcs_pack_patches launch in _forward_backbone)
        pe_in = ops.view(self._buf("pe_in", (B, H, H, 64)))
        e = self._tokens("pe.out", B, H, H, self.embed)
        self._lin(pe_in, PE, e)
        x = Act(self._tokens("pe.norm", B, H, H, self.embed))
        st0 = self._ln(e, "backbone.swin.embeddings.norm", x.v)
        if train:
            def pe_bwd(x=x, H=H):
                g = self._sum_grads(x, "pe.g")
                g_e = self._tokens("pe.g_e", B, H, H, self.embed)
                self._ln_bwd(e, g, "backbone.swin.embeddings.norm", st0, g_e)
                self._lin_bwd(pe_in, g_e, PE, None, want_dx=False)
                self._ready(PE + ".weight")
            tape.append(pe_bwd)
        # ---- UPerNet input buffers the encoder writes into
        psp_cat = self._act("psp_cat", B, h4, h4, self.dims[3] + len(POOL_SCALES) * HIDDEN)
        feats = []
        for s, (depth, heads) in enumerate(zip(self.swin_depths, self.swin_heads)):
            C_ = self.dims[s]
            for b in range(depth):
                p = f"backbone.swin.encoder.layers.{s}.blocks.{b}"
                x = self._block(x, p, heads, 0 if b % 2 == 0 else WINDOW // 2, self._tokens(p + ".out", B, H, H, C_), train, tape)
            fv = View(psp_cat, 0, self.dims[3]) if s == 3 else self._tokens(f"feat{s}", B, H, H, C_)
            f = Act(fv)
            nname = f"backbone.hidden_states_norms.stage{s + 1}"
            stn = self._ln(x.v, nname, f.v)
            feats.append(f)
            if train:
                def out_bwd(x=x, f=f, nname=nname, stn=stn, s=s):
                    g = self._sum_grads(f, f"feat{s}.g")
                    gx = self._tokens(f"feat{s}.gx", x.v.B, x.v.H, x.v.W, x.v.C)
                    self._ln_bwd(x.v, g, nname, stn, gx)
                    x.grads.append((gx, False))
                tape.append(out_bwd)
            if s < 3:
                d = f"backbone.swin.encoder.layers.{s}.downsample"
                m = self._tokens(d + ".merged", B, H // 2, H // 2, 4 * C_)
                ops.patch_merge(x.v, m)
                mn = self._tokens(d + ".normed", B, H // 2, H // 2, 4 * C_)
                stm = self._ln(m, d + ".norm", mn)
                nx = Act(self._tokens(d + ".out", B, H // 2, H // 2, 2 * C_))
                self._lin(mn, d + ".reduction", nx.v, bias=False)
                if train:
                    def ds_bwd(x=x, nx=nx, d=d, m=m, mn=mn, stm=stm, Hs=H, C_=C_):
                        g = self._sum_grads(nx, d + ".g")
                        g_mn = self._lin_bwd(mn, g, d + ".reduction", d + ".g_mn", bias=False)
                        g_m = self._tokens(d + ".g_m", B, Hs // 2, Hs // 2, 4 * C_)
                        self._ln_bwd(m, g_mn, d + ".norm", stm, g_m)
                        gx = self._tokens(d + ".gx", B, Hs, Hs, C_)
                        ops.patch_merge(gx, g_m, reverse=True)
                        x.grads.append((gx, False))
                        self._ready(d + ".reduction.weight")
                    tape.append(ds_bwd)
                x, H = nx, H // 2
        # ---- UPerNet head
        D = "decode_head"
        f4 = feats[3]
        psp_units = []
        for i, sc in enumerate(POOL_SCALES):
            pooled = self._tokens(f"psp{i}.pooled", B, sc, sc, self.dims[3])
            ops.adaptive_avg_pool(f4.v, pooled)
            pa = Act(self._tokens(f"psp{i}.act", B, sc, sc, HIDDEN))
            u = self._unit(pooled, f"{D}.psp_modules.{i}.1.conv", f"{D}.psp_modules.{i}.1.batch_norm", 1, 1, 0, train, pa.v)
            ops.resize_any(pa.v, View(psp_cat, self.dims[3] + i * HIDDEN, HIDDEN))
            psp_units.append((u, pooled, pa, sc))
        lat = [None] * 4
        lat_units = [None] * 4
        for i in range(3):
            hi = S // (4 * 2 ** i)
            lat[i] = Act(self._tokens(f"lat{i}", B, hi, hi, HIDDEN))
            lat_units[i] = self._unit(feats[i].v, f"{D}.lateral_convs.{i}.conv", f"{D}.lateral_convs.{i}.batch_norm", 1, 1, 0, train, lat[i].v)
        lat[3] = Act(self._tokens("lat3", B, h4, h4, HIDDEN))
        lat_units[3] = self._unit(ops.view(psp_cat), f"{D}.bottleneck.conv", f"{D}.bottleneck.batch_norm", 3, 1, 1, train, lat[3].v)
        ls = [None, None, None, lat[3]]                      # top-down sums
        ups = [None] * 3
        for i in range(3, 0, -1):
            hi = S // (4 * 2 ** (i - 1))
            ups[i - 1] = self._tokens(f"td{i - 1}.up", B, hi, hi, HIDDEN)
            ops.resize_any(ls[i].v, ups[i - 1])
            ls[i - 1] = Act(self._tokens(f"td{i - 1}.sum", B, hi, hi, HIDDEN))
            self._add(lat[i - 1].v, ups[i - 1], ls[i - 1].v)
        fcat = self._act("fpn_cat", B, h1, h1, 4 * HIDDEN)
        fpn, fpn_units = [None] * 4, [None] * 3
        for i in range(3):
            hi = S // (4 * 2 ** i)
            fpn[i] = Act(View(fcat, 0, HIDDEN) if i == 0 else self._tokens(f"fpn{i}", B, hi, hi, HIDDEN))
            fpn_units[i] = self._unit(ls[i].v, f"{D}.fpn_convs.{i}.conv", f"{D}.fpn_convs.{i}.batch_norm", 3, 1, 1, train, fpn[i].v)
        fpn[3] = ls[3]
        for i in range(1, 4):
            ops.resize_any(fpn[i].v, View(fcat, i * HIDDEN, HIDDEN))
        fb = Act(self._tokens("fpn_bottleneck", B, h1, h1, HIDDEN))
        ufb = self._unit(ops.view(fcat), f"{D}.fpn_bottleneck.conv", f"{D}.fpn_bottleneck.batch_norm", 3, 1, 1, train, fb.v)
        if train:
            def head_bwd():
                assert len(fb.grads) == 1
                dy = self._unit_bwd(ufb, fb.grads[0][0], 0)
                gcat, _ = self._dgrad(ufb, dy, "fpn_cat.g")
                g_fpn = [View(gcat.t, 0, HIDDEN)]
                for i in range(1, 4):
                    hi = S // (4 * 2 ** i)
                    gi = self._tokens(f"fpn{i}.g", B, hi, hi, HIDDEN)
                    ops.resize_any(View(gcat.t, i * HIDDEN, HIDDEN), gi, backward=True)
                    g_fpn.append(gi)
                g_ls = [None] * 4
                for i in range(3):
                    dyi = self._unit_bwd(fpn_units[i], g_fpn[i], 0)
                    g_ls[i] = self._dgrad(fpn_units[i], dyi, f"fpn{i}.g_ls")[0]
                g_ls[3] = g_fpn[3]
                # top-down transposes: ls[i-1] = lat[i-1] + up(ls[i])
                for i in range(1, 4):
                    hi = S // (4 * 2 ** i)
                    back = self._tokens(f"td{i}.g_back", B, hi, hi, HIDDEN)
                    ops.resize_any(g_ls[i - 1], back, backward=True)
                    tot = self._tokens(f"td{i}.g_tot", B, hi, hi, HIDDEN)
                    self._add(g_ls[i], back, tot)
                    g_ls[i] = tot
                for i in range(3):
                    dyl = self._unit_bwd(lat_units[i], g_ls[i], 0)
                    feats[i].grads.append(self._dgrad(lat_units[i], dyl, f"lat{i}.g_feat"))
                dyb = self._unit_bwd(lat_units[3], g_ls[3], 0)
                gpc, _ = self._dgrad(lat_units[3], dyb, "psp_cat.g")
                parts = [(View(gpc.t, 0, self.dims[3]), False)]
                for i, (u, pooled, pa, sc) in enumerate(psp_units):
                    g_pa = self._tokens(f"psp{i}.g_act", B, sc, sc, HIDDEN)
                    ops.resize_any(View(gpc.t, self.dims[3] + i * HIDDEN, HIDDEN), g_pa, backward=True)
                    dyp = self._unit_bwd(u, g_pa, 0)
                    g_pool, _ = self._dgrad(u, dyp, f"psp{i}.g_pooled")
                    g_f4 = self._tokens(f"psp{i}.g_f4", B, h4, h4, self.dims[3])
                    ops.adaptive_avg_pool(g_pool, g_f4, backward=True)
                    parts.append((g_f4, False))
                s3 = self._tokens("f4.g_sum3", B, h4, h4, self.dims[3])
                ops.relu_bwd_sum(None, parts[:3], s3)
                f4.grads.append((s3, False))
                f4.grads.extend(parts[3:])
                self._ready(f"{D}.psp_modules.0.1.conv.weight")
            tape.append(head_bwd)
        if train:
            self._tape = tape
        self._last = fb
        return fb

    def _plan(self, B, S):
        if self.shape != (B, S):
            assert S % 32 == 0, "tile side must be a multiple of 32"
            if self._switch_shape(B, S):
                self._rec, self._bufs, self.gn = {}, {}, {}
                self.logits = torch.empty((B, self.NC, S, S), dtype=torch.float32, device=self.dev)
                self.logits_lo = torch.empty((B, self.NC, S // 4, S // 4), dtype=torch.float32, device=self.dev)
                self.dlogits_lo = torch.empty_like(self.logits_lo)

    def _forward_backbone(self, x, train):
        B, C_, S, S2 = x.shape
        assert C_ == 3 and S == S2
        if train and B < 2:   # (torch raises the same for the BatchNorm of the 1x1 pyramid-pooling branch)
            raise ValueError("Expected more than 1 value per channel when training: Swin-T + UPerNet needs a batch of at least 2 tiles")
        self._plan(B, S)
        ops.pack_patches(x.contiguous(), self._buf("pe_in", (B, S // 4, S // 4, 64)))
        key = "fwd_train" if train else "fwd_eval"
        if train and "fwd_train" not in self._rec:
            self._rec.pop("bwd", None)
        self._run(key, lambda: self._backbone(B, S, train))
        if key not in self._last_act:
            self._last_act[key] = self._last
        self._saved_train = train
        return self._last_act[key]

    def forward(self, x: torch.Tensor, train: bool) -> torch.Tensor:
        h = self._forward_backbone(x, train)
        ops.linear_head_fwd(h.v, self.P[self.head_name + ".weight"].view(self.NC, HIDDEN), self.P[self.head_name + ".bias"], logits=self.logits_lo)
        ops.resize_bilinear_nchw_fwd(self.logits_lo, self.logits, 4)
        return self.logits

    def forward_labels(self, x: torch.Tensor, labels: torch.Tensor):
        ops.argmax_confusion(self.forward(x, False), labels)
        return labels

    def backward(self, dlogits: torch.Tensor):
        assert self._saved_train, "backward needs a preceding forward in train mode"
        if self.on_backward_begin is not None:
            self.on_backward_begin()
        B, S = self.shape
        NC = self.NC
        h = self._last_act["fwd_train"]
        ops.resize_bilinear_nchw_bwd(dlogits.contiguous(), self.dlogits_lo, 4)
        P = B * (S // 4) * (S // 4)
        rows = ops.linear_head_bwd_rows(P)
        part = self._scratch("head_part", rows * (NC * HIDDEN + NC))
        gh = ops.view(self._act("head.gx", B, S // 4, S // 4, HIDDEN))
        ops.linear_head_bwd(h.v, self.dlogits_lo, self.P[self.head_name + ".weight"].view(NC, HIDDEN), gh, part)
        gw, gb = self.G[self.head_name + ".weight"], self.G[self.head_name + ".bias"]
        assert gb.data_ptr() == gw.data_ptr() + gw.numel() * 4
        ops.colsum_finalize(part, rows, NC * HIDDEN + NC, gw)
        if self.on_grad_ready is not None:
            self.on_grad_ready(self._goff[self.head_name + ".weight"], [])

        def run_tape():
            h.grads = [(gh, False)]
            for fn in reversed(self._tape):
                fn()
            if self.fp8:
                self.f8.update()      # delayed scaling: this step's amax of every quantised tensor becomes the next step's scale
        self._run("bwd", run_tape)
