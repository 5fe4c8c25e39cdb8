"""Launch plan of the reference's `SegformerMod` (/root/reference/source/scripts/nets.py:313-349; factory name at
/root/reference/source/scripts/utils.py:191-192) on the HIP kernels: transformers' Mix-Transformer encoder + all-MLP decode head
(`SegformerForSemanticSegmentation`), then the reference's own ConvTranspose(8,2,3) / ConvTranspose(4,2,1) / Conv3x3 tail on the NC logits.

MI355X-first choices:
  * tokens ARE NHWC pixels (as in swin_engine.py): every nn.Linear is a 1x1 convolution on the GEMM kernels, key and value projections are
    ONE GEMM (their weights are adjacent in the flat parameter buffer);
  * every convolution that is neither 1x1 nor 3x3 / stride 1 - the overlapping patch embeddings (7/4/3, 3/2/1), the sequence-reduction
    convolutions (k = s = 8 | 4 | 2) - is ONE patch gather (`cvcs_im2col`) + the same GEMM kernels (forward, data gradient, weight
    gradient); each ConvTranspose2d(k, 2, p) of the tail is ONE stride-1 convolution to 4 x the channels over a (k/2 + 1)^2 window + a pixel
    shuffle (`cvcs_deconv_pack`): k = 4 on the 3x3 conv kernels, k = 8 as a 5x5 patch GEMM; the reference's input normalisation is folded into the first gather;
  * the master parameters keep the reference's shapes (Conv2d OIHW, ConvTranspose2d IOHW, NC channels): the GEMM operands (K-major patch
    order, channels padded to 32) are re-gathered from them each step by one table-driven launch, their gradients scattered back;
  * attention never materialises the N x Nk probabilities: the forward keeps the log-sum-exp, the backward recomputes;
  * DropPath is a per-sample factor drawn by a counter-based generator (replayable on the host) and fused into the residual adds; the
    head's Dropout is `cvcs_dropout`;
  * the NC-channel tail runs NHWC with channels padded to 32; the NCHW f32 logits leave through a
    transposition kernel (`cvcs_planes_from_nhwc`), the gradient comes back the same way;
  * recorded launch plans as in resnet_engine.py.
"""
from __future__ import annotations

import torch

from . import _lib, ops
from .ops import View
from .resnet_engine import Act
from .swin_engine import SwinUPerNetEngine

CONFIGS = {
    "b0": dict(hidden=(32, 64, 160, 256), depths=(2, 2, 2, 2), heads=(1, 2, 5, 8), sr=(8, 4, 2, 1), patch=(7, 3, 3, 3), strides=(4, 2, 2, 2),
               mlp_ratio=4, decoder=256, drop_path=0.1, classifier_dropout=0.1),
    "b3": dict(hidden=(64, 128, 320, 512), depths=(3, 4, 18, 3), heads=(1, 2, 5, 8), sr=(8, 4, 2, 1), patch=(7, 3, 3, 3), strides=(4, 2, 2, 2),
               mlp_ratio=4, decoder=768, drop_path=0.1, classifier_dropout=0.1),
}
ENC, HEAD = "segformer.segformer.stages", "segformer.decode_head"
BIG = 1 << 20


def drop_path_rates(cfg):
    n = sum(cfg["depths"])
    return [cfg["drop_path"] * i / max(n - 1, 1) for i in range(n)]


def _pad32(n):
    return -(-n // 32) * 32


class SegformerEngine(SwinUPerNetEngine):
    def __init__(self, num_classes: int, dtype: torch.dtype, device, variant="b0", stochastic=True, seed=0):
        super().__init__(num_classes, dtype, device)
        self.cfg = CONFIGS[variant]
        self.variant = variant
        assert all(c // h in (32, 64) for c, h in zip(self.cfg["hidden"], self.cfg["heads"])), "the attention kernels are built for head dimensions 32 and 64"
        self.head_name = "seq.4"
        assert num_classes <= 32, "SegformerMod's NC-channel tail is built for at most 32 classes (the head kernels' limit)"
        self.CP = 32          # channel count of the NC-channel maps (the 1x1 GEMM kernels write multiples of 32 output channels)
        # train-mode DropPath / Dropout (the library's defaults, active under net.train() at S/train.py:113); off = deterministic parity runs
        self.stochastic = bool(stochastic)
        self.drop_state = torch.tensor([seed, 0], dtype=torch.int64, device=self.dev)
        self.rates = drop_path_rates(self.cfg)

    # ------------------------------------------------------------------------------------------------ binding
    def bind(self, params, grads, buffers):
        self.P, self.G, self.Bf = params, grads, buffers
        base = min(g.data_ptr() for g in grads.values())
        self._goff = {k: (g.data_ptr() - base) // 4 for k, g in grads.items()}
        dev, dt, cfg, NC, CP = self.dev, self.dtype, self.cfg, self.NC, self.CP
        kg = ops.KGROUP[ops.dtype_code(dt)]
        self.packed, self.W4, self.GW4 = {}, {}, {}
        entries, gath, scat = [], [], []
        self._scat_index = {}

        def plain(key, w4, g4):
            cout, cin, kh, kw = w4.shape
            pk = dict(wf=torch.empty(kh * kw, cout, cin, dtype=dt, device=dev), wd=torch.empty(kh * kw, cin, cout, dtype=dt, device=dev))
            self.packed[key], self.W4[key], self.GW4[key] = pk, w4, g4
            entries.append((w4, pk["wf"], pk["wd"]))

        def gathered(key, wname, R, Rv, K, row, col, base=0, rv2=0, cv2=0, bias=None, bias_key=None):
            """a GEMM layer whose operands are gathered from the master tensor `wname`: [R][K] = (row index, column index) -> master element"""
            Kp = _pad32(K)
            w, g = params[wname], grads[wname]
            cw = max(R, kg)
            wf = torch.zeros(1, R, Kp, dtype=dt, device=dev)
            wd = torch.zeros(1, Kp, cw, dtype=dt, device=dev)
            gw = torch.zeros(R, Kp, dtype=torch.float32, device=dev)
            gath.append(dict(src=w, dst=wf, base=base, R=R, Cp=Kp, Rv=Rv, Cv=K, row=row, col=col, rv2=rv2, cv2=cv2))
            gath.append(dict(src=w, dst=wd, base=base, R=Kp, Cp=cw, Rv=K, Cv=Rv, row=col, col=row, rv2=cv2, cv2=rv2))
            self._scat_index[key] = len(scat)
            scat.append(dict(src=g, dst=gw, base=base, R=R, Cp=Kp, Rv=Rv, Cv=K, row=row, col=col, rv2=rv2, cv2=cv2, f32_out=True))
            self.GW4[key] = gw.view(R, Kp, 1, 1)
            self.packed[key] = dict(wf=wf, wd=wd)
            bk = bias_key or key
            if bias is not None and (bk + ".bias") not in self.W4:
                Rb = R
                bp = torch.zeros(Rb, dtype=torch.float32, device=dev)
                gb = torch.zeros(Rb, dtype=torch.float32, device=dev)
                gath.append(dict(src=params[bias], dst=bp, R=1, Cp=Rb, Rv=1, Cv=params[bias].numel(), row=(1, BIG, 0, 0, 0), col=(1, BIG, 0, 0, 1), f32_out=True))
                self._scat_index[bk + ".bias"] = len(scat)
                scat.append(dict(src=grads[bias], dst=gb, R=1, Cp=Rb, Rv=1, Cv=params[bias].numel(), row=(1, BIG, 0, 0, 0), col=(1, BIG, 0, 0, 1), f32_out=True))
                self.W4[bk + ".bias"], self.GW4[bk + ".bias"] = bp, gb
            if bias is not None:
                self.W4[key + ".bias"], self.GW4[key + ".bias"] = self.W4[bk + ".bias"], self.GW4[bk + ".bias"]

        def conv_as_gemm(key, cout_pad=None):
            """nn.Conv2d [Co, Ci, k, k] as a GEMM over (ky, kx, ci) patches"""
            w = params[key + ".weight"]
            co, ci, k, _ = w.shape
            T = k * k
            gathered(key, key + ".weight", cout_pad or co, co, T * ci, row=(1, BIG, 0, 0, ci * T), col=(1, ci, 1, 0, T), bias=key + ".bias")

        cin = 3
        for s, (C_, depth, sr) in enumerate(zip(cfg["hidden"], cfg["depths"], cfg["sr"])):
            conv_as_gemm(f"{ENC}.{s}.patch_embeddings.proj")
            for b in range(depth):
                p = f"{ENC}.{s}.blocks.{b}"
                att = p + ".attention"
                for n_ in ("q_proj", "o_proj"):
                    w = params[f"{att}.{n_}.weight"]
                    plain(f"{att}.{n_}", w.view(C_, C_, 1, 1), grads[f"{att}.{n_}.weight"].view(C_, C_, 1, 1))
                kw_, vw = params[att + ".k_proj.weight"], params[att + ".v_proj.weight"]
                assert vw.data_ptr() == kw_.data_ptr() + C_ * C_ * 4 and params[att + ".v_proj.bias"].data_ptr() == params[att + ".k_proj.bias"].data_ptr() + C_ * 4, \
                    "key / value projections must be adjacent in the flat buffers (nets.SegformerMod._flat_order)"
                fuse = lambda t, shape: torch.as_strided(t, shape, tuple(int(torch.tensor(shape[i + 1:]).prod()) for i in range(len(shape))))  # noqa: E731
                plain(att + ".kv", fuse(kw_, (2 * C_, C_, 1, 1)), fuse(grads[att + ".k_proj.weight"], (2 * C_, C_, 1, 1)))
                self.W4[att + ".kv.bias"] = fuse(params[att + ".k_proj.bias"], (2 * C_,))
                self.GW4[att + ".kv.bias"] = fuse(grads[att + ".k_proj.bias"], (2 * C_,))
                if sr > 1:
                    conv_as_gemm(att + ".sequence_reduction.sequence_reduction")
                for n_ in ("fc1", "fc2"):
                    w = params[f"{p}.mlp.{n_}.weight"]
                    plain(f"{p}.mlp.{n_}", w.view(w.shape[0], w.shape[1], 1, 1), grads[f"{p}.mlp.{n_}.weight"].view(w.shape[0], w.shape[1], 1, 1))
                gw, gb = grads[p + ".mlp.dwconv.dwconv.weight"], grads[p + ".mlp.dwconv.dwconv.bias"]
                assert gb.data_ptr() == gw.data_ptr() + gw.numel() * 4, "depthwise weight | bias gradients must be adjacent"
            cin = C_
        D = cfg["decoder"]
        for s, C_ in enumerate(cfg["hidden"]):
            k_ = f"{HEAD}.linear_projections.{s}.proj"
            plain(k_, params[k_ + ".weight"].view(D, C_, 1, 1), grads[k_ + ".weight"].view(D, C_, 1, 1))
        plain(HEAD + ".linear_fuse", params[HEAD + ".linear_fuse.weight"], grads[HEAD + ".linear_fuse.weight"])
        # ---- the NC-channel layers: channels padded to CP, pad rows / columns zero
        gathered(HEAD + ".classifier", HEAD + ".classifier.weight", CP, NC, D, row=(1, BIG, 0, 0, D), col=(1, BIG, 0, 0, 1), bias=HEAD + ".classifier.bias")
        # the two ConvTranspose2d of the tail, each as ONE stride-1 convolution CP -> 4 CP (cvcs_deconv_pack): seq.2 (k = 4) on the 3x3 conv
        # kernels, seq.0 (k = 8, a 5x5 window) as a patch GEMM
        self.deconv = {}
        for name, k, pd, layout in (("seq.0", 8, 3, 1), ("seq.2", 4, 1, 0)):
            T = k // 2 + 1
            if layout == 0:
                wf = torch.zeros(T * T, 4 * CP, max(CP, kg), dtype=dt, device=dev)
                wd = torch.zeros(T * T, CP, 4 * CP, dtype=dt, device=dev)
                gw = torch.zeros(4 * CP, CP, T, T, dtype=torch.float32, device=dev)
            else:
                Kp = -(-T * T * CP // 128) * 128          # (a multiple of 128: the data-gradient GEMM writes Kp columns on the taps kernel)
                wf = torch.zeros(1, 4 * CP, Kp, dtype=dt, device=dev)
                wd = torch.zeros(1, Kp, 4 * CP, dtype=dt, device=dev)
                gw = torch.zeros(4 * CP, Kp, 1, 1, dtype=torch.float32, device=dev)
            self.deconv[name] = dict(k=k, p=pd, T=T, layout=layout, wf=wf, wd=wd, gw=gw, bias4=torch.zeros(4 * CP, dtype=torch.float32, device=dev),
                                     gb4=torch.zeros(4 * CP, dtype=torch.float32, device=dev))
            self.packed[name] = dict(wf=wf, wd=wd)
            self.W4[name + ".bias"], self.GW4[name] = self.deconv[name]["bias4"], gw
            self.GW4[name + ".bias"] = self.deconv[name]["gb4"]
        # the last 3x3 (NC -> NC) stays on the conv kernels: [tap][CP][CPk] forward, flipped [tap][CP][CPk] data gradient, OIHW-padded gradient
        cpk = max(CP, kg)
        w3, g3 = params["seq.4.weight"], grads["seq.4.weight"]
        wf3 = torch.zeros(9, CP, cpk, dtype=dt, device=dev)
        wd3 = torch.zeros(9, CP, cpk, dtype=dt, device=dev)
        gw3 = torch.zeros(CP, CP, 3, 3, dtype=torch.float32, device=dev)
        gath.append(dict(src=w3, dst=wf3, R=9 * CP, Cp=cpk, Rv=9 * CP, Cv=NC, row=(1, CP, 1, 0, NC * 9), col=(1, BIG, 0, 0, 9), rv2=NC))
        gath.append(dict(src=w3, dst=wd3, base=8, R=9 * CP, Cp=cpk, Rv=9 * CP, Cv=NC, row=(1, CP, -1, 0, 9), col=(1, BIG, 0, 0, NC * 9), rv2=NC))
        self._scat_index["seq.4"] = len(scat)
        scat.append(dict(src=g3, dst=gw3, R=CP, Cp=CP * 9, Rv=NC, Cv=NC * 9, row=(1, BIG, 0, 0, NC * 9), col=(1, BIG, 0, 0, 1), f32_out=True))
        self.packed["seq.4"] = dict(wf=wf3, wd=wd3)
        self.GW4["seq.4"] = gw3
        bp, gb = torch.zeros(CP, dtype=torch.float32, device=dev), torch.zeros(CP, dtype=torch.float32, device=dev)
        gath.append(dict(src=params["seq.4.bias"], dst=bp, R=1, Cp=CP, Rv=1, Cv=NC, row=(1, BIG, 0, 0, 0), col=(1, BIG, 0, 0, 1), f32_out=True))
        self._scat_index["seq.4.bias"] = len(scat)
        scat.append(dict(src=grads["seq.4.bias"], dst=gb, R=1, Cp=CP, Rv=1, Cv=NC, row=(1, BIG, 0, 0, 0), col=(1, BIG, 0, 0, 1), f32_out=True))
        self.W4["seq.4.bias"], self.GW4["seq.4.bias"] = bp, gb
        self._pack_table = ops.pack_table(entries, dev)
        self._gather = ops.gather_table(gath, dev)
        self._scatter = ops.gather_table(scat, dev)
        self._keep = (gath, scat)           # (the tables hold raw pointers into these tensors)
        from .engine import _BN
        self.bn = {HEAD + ".batch_norm": _BN(D, dev)}
        self.one = torch.ones(4096, dtype=torch.float32, device=dev)
        self.zero = torch.zeros(4096, dtype=torch.float32, device=dev)
        self.eye = torch.eye(NC, CP, dtype=torch.float32, device=dev).contiguous()
        self.rates_dev = torch.tensor([r for r in self.rates for _ in (0, 1)], dtype=torch.float32, device=dev)
        self._rec, self._bufs, self.shape, self.gn = {}, {}, None, {}
        self._shape_cache.clear()

    def refresh_weights(self, train=True):
        ops.pack_conv_weights(self._pack_table[0], self._pack_table[1], self.dtype)
        ops.gather_weights(self._gather[0], self._gather[1], self.dtype)
        for name, d in self.deconv.items():
            ops.deconv_pack(self.P[name + ".weight"], self.P[name + ".bias"], d["k"], d["p"], self.CP, d["layout"], d["wf"], d["wd"], d["bias4"])

    def _scatter_grad(self, key):
        ops.scatter_weight_grads(self._scatter[0], 1, first=self._scat_index[key])

    # ------------------------------------------------------------------------------------------------ pieces
    def _lin(self, x: View, key, out: View, bias=True):
        ops.SCOPE = "enc" if key.startswith(ENC) else "dec"
        b = None
        if bias:
            b = self.W4.get(key + ".bias")
            if b is None:
                b = self.P[key + ".bias"]
        ops.conv2d(x, self.packed[key]["wf"], b, out, 1, 1)

    def _gemm_bwd(self, x: View, g: View, key, gx_name, bias=True, want_dx=True, scatter=True):
        """_lin_bwd of a gathered layer + the scatter of its gradient into the master layout"""
        gx = self._lin_bwd(x, g, key, gx_name, bias=bias, want_dx=want_dx)
        if scatter:
            self._scatter_grad(key)
            if bias:
                self._scatter_grad(key + ".bias")
        return gx

    def _patch_conv(self, x: View | None, key, k, stride, pad, out_name, train, tape, xa: Act | None):
        """nn.Conv2d(k, stride, pad) with bias as a patch gather + GEMM.  x None: the first patch embedding (its patch matrix is written by
        the eager cvcs_im2col_stem launch in _forward_backbone).  Returns the output Act"""
        B = self.shape[0]
        cout = self.packed[key]["wf"].shape[1]
        Kp = self.packed[key]["wf"].shape[2]
        if x is None:
            Ho = Wo = self.shape[1] // stride
            col = ops.view(self._buf("stem_col", (B, Ho, Wo, Kp)))
        else:
            Ho, Wo = ops.conv_out_hw(x.H, x.W, k, k, stride, pad)
            col = ops.view(self._act(key + ".col", B, Ho, Wo, Kp))
            ops.im2col(x, col, k, k, stride, -pad, -pad)
        out = Act(self._tokens(out_name, B, Ho, Wo, cout))
        self._lin(col, key, out.v)
        if train:
            def bwd():
                g = self._sum_grads(out, out_name + ".g")
                if x is None:
                    self._gemm_bwd(col, g, key, None, want_dx=False)
                else:
                    gcol = self._gemm_bwd(col, g, key, key + ".gcol")
                    gx = self._tokens(key + ".gx", B, x.H, x.W, x.C)
                    ops.col2im(gcol, gx, k, k, stride, -pad, -pad)
                    xa.grads.append((gx, False))
            tape.append(bwd)
        return out

    def _ln_act(self, x: Act, name, out_name, train, tape, out_view: View | None = None) -> Act:
        v = x.v
        o = Act(out_view if out_view is not None else self._tokens(out_name, v.B, v.H, v.W, v.C))
        st = self._ln(v, name, o.v)
        if train:
            def bwd():
                g = self._sum_grads(o, out_name + ".g")
                gx = self._tokens(out_name + ".gx", v.B, v.H, v.W, v.C)
                self._ln_bwd(v, g, name, st, gx)
                x.grads.append((gx, False))
            tape.append(bwd)
        return o

    def _residual(self, x: Act, branch: Act, call, out_name, train, tape) -> Act:
        """out = x + DropPath(branch): `call` = row of self.dp_scales (None: no DropPath on this block)"""
        v = x.v
        o = Act(self._tokens(out_name, v.B, v.H, v.W, v.C))
        sc = self.dp_scales[call] if (train and call is not None) else None
        ops.scale_rows_add(branch.v, sc, v, o.v)
        if train:
            def bwd():
                g = self._sum_grads(o, out_name + ".g")
                x.grads.append((g, False))
                if sc is None:
                    branch.grads.append((g, False))
                else:
                    gb = self._tokens(out_name + ".gbr", v.B, v.H, v.W, v.C)
                    ops.scale_rows_add(g, sc, None, gb)
                    branch.grads.append((gb, False))
            tape.append(bwd)
        return o

    def _attention(self, h: Act, p, heads, sr, train, tape) -> Act:
        """SegformerAttention on the LayerNormed tokens h -> the output projection"""
        att = p + ".attention"
        v = h.v
        B, H, W, C_ = v.B, v.H, v.W, v.C
        q = Act(self._tokens(p + ".q", B, H, W, C_))
        self._lin(v, att + ".q_proj", q.v)
        if sr > 1:
            red = self._patch_conv(v, att + ".sequence_reduction.sequence_reduction", sr, sr, 0, p + ".sr", train, tape, h)
            kvin = self._ln_act(red, att + ".sequence_reduction.layer_norm", p + ".sr_ln", train, tape)
        else:
            kvin = h
        Hk, Wk = kvin.v.H, kvin.v.W
        kv = Act(self._tokens(p + ".kv", B, Hk, Wk, 2 * C_))
        self._lin(kvin.v, att + ".kv", kv.v)
        ao = Act(self._tokens(p + ".ao", B, H, W, C_))
        lse = self._buf(p + ".lse", (B * heads * H * W,), torch.float32)
        ops.SCOPE = "enc"
        ops.sr_attention_fwd(q.v, kv.v, heads, ao.v, lse)
        pr = Act(self._tokens(p + ".proj", B, H, W, C_))
        self._lin(ao.v, att + ".o_proj", pr.v)
        if train:
            def bwd():
                g = self._sum_grads(pr, p + ".proj.g")
                g_ao = self._lin_bwd(ao.v, g, att + ".o_proj", p + ".g_ao")
                gq = self._tokens(p + ".gq", B, H, W, C_)
                gkv = self._tokens(p + ".gkv", B, Hk, Wk, 2 * C_)
                ws = self._scratch("sr_attn_ws", ops.sr_attention_bwd_workspace(B, H * W, Hk * Wk, heads, C_ // heads))
                ops.SCOPE = "enc"
                ops.sr_attention_bwd(q.v, kv.v, ao.v, g_ao, lse, heads, gq, gkv, ws)
                g_kvin = self._lin_bwd(kvin.v, gkv, att + ".kv", p + ".g_kvin")
                kvin.grads.append((g_kvin, False))
                g_h = self._lin_bwd(v, gq, att + ".q_proj", p + ".g_hq")
                h.grads.append((g_h, False))
            # (registered BEFORE the closures of the reduction branch would be wrong: the tape runs in reverse, so this closure - appended last -
            # runs first and feeds kvin.grads / h.grads before the reduction's closures consume them)
            tape.append(bwd)
        return pr

    def _mix_ffn(self, h: Act, p, train, tape) -> Act:
        v = h.v
        B, H, W, C_ = v.B, v.H, v.W, v.C
        Ch = self.cfg["mlp_ratio"] * C_
        f1 = self._tokens(p + ".f1", B, H, W, Ch)
        self._lin(v, p + ".mlp.fc1", f1)
        dw = self._tokens(p + ".dw", B, H, W, Ch)
        wdw, bdw = self.P[p + ".mlp.dwconv.dwconv.weight"], self.P[p + ".mlp.dwconv.dwconv.bias"]
        ops.SCOPE = "enc"
        ops.dwconv3x3(f1, wdw, bdw, dw)
        a = self._tokens(p + ".gelu", B, H, W, Ch)
        ops.gelu(dw, a)
        f2 = Act(self._tokens(p + ".f2", B, H, W, C_))
        self._lin(a, p + ".mlp.fc2", f2.v)
        if train:
            def bwd():
                g = self._sum_grads(f2, p + ".f2.g")
                g_a = self._lin_bwd(a, g, p + ".mlp.fc2", p + ".g_a")
                g_dw = self._tokens(p + ".g_dw", B, H, W, Ch)
                ops.gelu(dw, g_dw, g=g_a)
                rows = ops.dwconv3x3_wgrad_rows(B * H * W)
                part = self._scratch("dw_part", rows * 10 * Ch)
                ops.SCOPE = "enc"
                ops.dwconv3x3_wgrad(f1, g_dw, part)
                ops.colsum_finalize(part, rows, 10 * Ch, self.G[p + ".mlp.dwconv.dwconv.weight"])
                g_f1 = self._tokens(p + ".g_f1", B, H, W, Ch)
                ops.dwconv3x3(g_dw, wdw, None, g_f1, flip=True)
                g_h = self._lin_bwd(v, g_f1, p + ".mlp.fc1", p + ".g_hf")
                h.grads.append((g_h, False))
            tape.append(bwd)
        return f2

    # ------------------------------------------------------------------------------------------------ the network
    def _backbone(self, B, S, train):
        cfg, NC, CP = self.cfg, self.NC, self.CP
        tape = []
        self.units, self.relu_order = {}, []
        self.refresh_weights(train)
        n_calls = 2 * sum(cfg["depths"])
        stoch = train and self.stochastic
        if stoch:
            self.dp_scales = self._buf("dp_scales", (n_calls, B), torch.float32)
            ops.drop_path_scales(self.drop_state, self.rates_dev, B, self.dp_scales)
        feats = []
        x = None
        li = 0
        for s, (C_, depth, heads, sr, k, stride) in enumerate(zip(cfg["hidden"], cfg["depths"], cfg["heads"], cfg["sr"], cfg["patch"], cfg["strides"])):
            pre = f"{ENC}.{s}"
            if train:
                # (first closure of the stage = the last one its backward runs: every gradient at a flat offset >= this stage's first parameter
                # is final then - the signal the data-parallel reducer buckets on)
                tape.append(lambda pre=pre: self._ready(pre + ".patch_embeddings.proj.weight"))
            e = self._patch_conv(None if s == 0 else x.v, pre + ".patch_embeddings.proj", k, stride, k // 2, pre + ".pe", train, tape, x)
            x = self._ln_act(e, pre + ".patch_embeddings.layer_norm", pre + ".pe_ln", train, tape)
            for b in range(depth):
                p = f"{pre}.blocks.{b}"
                call = 2 * li if (stoch and self.rates[li] > 0) else None
                h1 = self._ln_act(x, p + ".layernorm_before", p + ".h1", train, tape)
                a = self._attention(h1, p, heads, sr, train, tape)
                x1 = self._residual(x, a, call, p + ".x1", train, tape)
                h2 = self._ln_act(x1, p + ".layernorm_after", p + ".h2", train, tape)
                f = self._mix_ffn(h2, p, train, tape)
                x = self._residual(x1, f, None if call is None else call + 1, p + ".out", train, tape)
                li += 1
            x = self._ln_act(x, pre + ".layer_norm", pre + ".feat", train, tape)
            feats.append(x)
        # ---- all-MLP decode head
        D = cfg["decoder"]
        h1 = S // 4
        cat = self._act("head.cat", B, h1, h1, 4 * D)
        projs = []
        for s, f in enumerate(feats):
            key = f"{HEAD}.linear_projections.{s}.proj"
            pv = self._tokens(f"head.proj{s}", B, f.v.H, f.v.W, D)
            self._lin(f.v, key, pv)
            slot = View(cat, (3 - s) * D, D)                 # torch.cat(all_hidden_states[::-1]): the last stage first
            ops.SCOPE = "dec"
            if s == 0:
                ops.scale_rows_add(pv, None, None, slot)     # (stage 1 is already at the target size: a copy into its channel range)
            else:
                ops.resize_any(pv, slot)
            projs.append(pv)
        fa = Act(self._tokens("head.fused", B, h1, h1, D))
        uf = self._unit(ops.view(cat), HEAD + ".linear_fuse", HEAD + ".batch_norm", 1, 1, 0, train, fa.v)
        p_drop = cfg["classifier_dropout"]
        if stoch and p_drop > 0:
            ops.dropout(fa.v, fa.v, self.drop_state, p_drop)
        z = Act(self._tokens("head.z", B, h1, h1, CP))
        self._lin(fa.v, HEAD + ".classifier", z.v)
        if train:
            def head_bwd():
                # classifier, dropout, linear_fuse + BatchNorm + ReLU, the four projections
                gz = self._sum_grads(z, "head.z.g")
                g_fa = self._gemm_bwd(fa.v, gz, HEAD + ".classifier", "head.g_fa")
                if stoch and p_drop > 0:
                    ops.dropout(g_fa, g_fa, self.drop_state, p_drop)
                dy = self._unit_bwd(uf, g_fa, 0)
                gcat, _ = self._dgrad(uf, dy, "head.cat.g")
                for s, f in enumerate(feats):
                    key = f"{HEAD}.linear_projections.{s}.proj"
                    slot = View(gcat.t, (3 - s) * D, D)
                    if s == 0:
                        gp = slot
                    else:
                        gp = self._tokens(f"head.proj{s}.g", B, f.v.H, f.v.W, D)
                        ops.SCOPE = "dec"
                        ops.resize_any(slot, gp, backward=True)
                    f.grads.append((self._lin_bwd(f.v, gp, key, f"head.feat{s}.g"), False))
                self._ready(HEAD + ".linear_projections.0.proj.weight")
            tape.append(head_bwd)
        # ---- the reference's tail on the NC logits (S/nets.py:331-335)
        t1 = self._deconv(z, "seq.0", 8, 3, train, tape)
        t2 = self._deconv(t1, "seq.2", 4, 1, train, tape)
        out = Act(self._tokens("seq.out", B, S, S, CP))
        ops.SCOPE = "dec"
        ops.conv2d(t2.v, self.packed["seq.4"]["wf"], self.W4["seq.4.bias"], out.v, 3, 3, 1, 1)
        if train:
            def last_bwd():
                g = self._sum_grads(out, "seq.out.g")
                ops.SCOPE = "dec"
                need = ops.wgrad_workspace_floats_for(t2.v, g, 3, 3, 1, 1)
                ops.conv2d_wgrad(t2.v, g, self.GW4["seq.4"], 3, 3, 1, 1, self._scratch("wg_ws", need))
                self._bias_grad(g, "seq.4")
                self._scatter_grad("seq.4")
                gt2 = self._tokens("seq.4.gx", B, S, S, CP)
                ops.conv2d(g, self.packed["seq.4"]["wd"], None, gt2, 3, 3, 1, 1)
                t2.grads.append((gt2, False))
            tape.append(last_bwd)
        if train:
            self._tape = tape
        self._last = out
        return out

    def _bias_grad(self, g: View, key):
        M = g.B * g.H * g.W
        rows = ops.bn_bwd_rows(M)
        part = self._scratch("bias_part", rows * g.C)
        ops.colsum_partial(g, part)
        ops.colsum_finalize(part, rows, g.C, self.GW4[key + ".bias"])
        self._scatter_grad(key + ".bias")

    def _deconv(self, x: Act, name, k, pd, train, tape) -> Act:
        """nn.ConvTranspose2d(NC, NC, k, stride 2, padding pd) + ReLU on a CP-channel map as ONE stride-1 convolution to 4 CP channels (phase-major)
        + the pixel shuffle: k = 4 on the 3x3 conv kernels, k = 8 (5x5 window) as a patch GEMM"""
        d = self.deconv[name]
        v = x.v
        B, H, W, CP = v.B, v.H, v.W, v.C
        T, layout = d["T"], d["layout"]
        y4 = self._tokens(name + ".y4", B, H, W, 4 * CP)
        ops.SCOPE = "dec"
        if layout == 0:
            col = None
            ops.conv2d(v, d["wf"], d["bias4"], y4, T, T, 1, T // 2)
        else:
            col = ops.view(self._act(name + ".col", B, H, W, d["wf"].shape[2]))
            ops.im2col(v, col, T, T, 1, -(T // 2), -(T // 2))
            ops.conv2d(col, d["wf"], d["bias4"], y4, 1, 1)
        o = Act(self._tokens(name + ".out", B, 2 * H, 2 * W, CP))
        ops.phase_shuffle(y4, CP, o.v, relu=True)
        if train:
            def bwd():
                g = self._sum_grads(o, name + ".out.g")
                g4 = self._tokens(name + ".g4", B, H, W, 4 * CP)
                ops.SCOPE = "dec"
                ops.phase_unshuffle(g, o.v, g4, CP)
                rows = ops.bn_bwd_rows(B * H * W)
                part = self._scratch("bias_part", rows * 4 * CP)
                ops.colsum_partial(g4, part)
                ops.colsum_finalize(part, rows, 4 * CP, d["gb4"])
                gx = self._tokens(name + ".gx", B, H, W, CP)
                if layout == 0:
                    need = ops.wgrad_workspace_floats_for(v, g4, T, T, 1, T // 2)
                    ops.conv2d_wgrad(v, g4, d["gw"], T, T, 1, T // 2, self._scratch("wg_ws", need))
                    ops.conv2d(g4, d["wd"], None, gx, T, T, 1, T // 2)
                else:
                    need = ops.wgrad_workspace_floats_for(col, g4, 1, 1, 1, 0)
                    ops.conv2d_wgrad(col, g4, d["gw"], 1, 1, 1, 0, self._scratch("wg_ws", need))
                    gcol = ops.view(self._act(name + ".gcol", B, H, W, col.C))
                    ops.conv2d(g4, d["wd"], None, gcol, 1, 1)
                    ops.col2im(gcol, gx, T, T, 1, -(T // 2), -(T // 2))
                ops.deconv_unpack_grad(d["gw"], d["gb4"], k, pd, CP, layout, self.G[name + ".weight"], self.G[name + ".bias"])
                x.grads.append((gx, False))
            tape.append(bwd)
        return o

    # ------------------------------------------------------------------------------------------------ passes
    def _plan(self, B, S):
        if self.shape != (B, S):
            assert S % 32 == 0, "tile side must be a multiple of 32"
            if self._switch_shape(B, S):
                self._rec, self._bufs, self.gn = {}, {}, {}
                self.logits = torch.empty((B, self.NC, S, S), dtype=torch.float32, device=self.dev)

    def _forward_backbone(self, x, train):
        B, C_, S, S2 = x.shape
        assert C_ == 3 and S == S2
        if train and B * (S // 4) ** 2 < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        self._plan(B, S)
        Kp = self.packed[f"{ENC}.0.patch_embeddings.proj"]["wf"].shape[2]
        k, st = self.cfg["patch"][0], self.cfg["strides"][0]
        ops.im2col_stem(x.contiguous(), ops.view(self._buf("stem_col", (B, S // st, S // st, Kp))), k, k, st, k // 2)
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
        ops.planes_from_nhwc(h.v, self.NC, self.logits)
        return self.logits

    def forward_labels(self, x: torch.Tensor, labels: torch.Tensor):
        h = self._forward_backbone(x, False)
        ops.head_argmax(h.v, self.eye, self.zero[:self.NC], labels)
        return labels

    def backward(self, dlogits: torch.Tensor):
        assert self._saved_train, "backward needs a preceding forward in train mode"
        if self.on_backward_begin is not None:
            self.on_backward_begin()
        B, S = self.shape
        NC, CP = self.NC, self.CP
        h = self._last_act["fwd_train"]
        gh = ops.view(self._act("head.gx", B, S, S, CP))
        ops.nhwc_from_planes(dlogits.contiguous(), NC, gh)

        def run_tape():
            h.grads = [(gh, False)]
            for fn in reversed(self._tape):
                fn()
            if self.stochastic:
                ops.counter_add(self.drop_state, 1)
        self._run("bwd", run_tape)
