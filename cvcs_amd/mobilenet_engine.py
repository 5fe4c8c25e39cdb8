"""Launch plan of the reference's factory name `MobileNet` (/root/reference/source/scripts/nets.py:277-311: torchvision's
`deeplabv3_mobilenet_v3_large` with `classifier[4]` -> Conv2d(256, num_classes, 1), forward returns d['out']; factory at
/root/reference/source/scripts/utils.py:183-184) on the HIP kernels.

Backbone = `mobilenet_v3_large(dilated=True).features` (output stride 16: the last three blocks trade their stride for dilation 2), head =
torchvision's DeepLabHead(960, NC) = the ASPP (rates 12/24/36) / 3x3 / classifier plan of deeplab_engine.py, which this engine inherits.

MI355X-first choices (the network is HBM-bound from end to end: 0.5 GFLOP per 512^2 tile in the backbone against ~0.25 GB of activations):
  * NHWC; every 1x1 convolution (expand, project, squeeze-excite FCs, the 960-wide last layer) is a GEMM on the existing kernels; the 3x3 /
    stride 2 stem is ONE patch gather straight from the planar u8 / f32 tile (`cvcs_im2col_stem`, no normalisation: S/nets.py:300-304 feeds
    the raw values) + the same GEMM;
  * channel widths are padded to multiples of 32 inside the engine (16/24 -> 32, 40 -> 64, 72/80 -> 96, 112/120 -> 128, 184 -> 192, 200 -> 224,
    240 -> 256): the master parameters keep torchvision's shapes, the padded operands (GEMM weights K-major, depthwise filters tap-major, the
    BatchNorm vectors and running statistics) are re-gathered from them every step by ONE table-driven launch and the gradients scattered
    back by one more; padding channels hold exact zeros through forward and backward (zero weights, gamma = beta = 0);
  * depthwise k x k (k = 3 | 5, stride 1 | 2, dilation 1 | 2) on `cvcs_dwconv*` (csrc/mobilenet.hip): one pass each for forward (BatchNorm
    statistics in its epilogue), data gradient and weight-gradient partials; hardswish is an activation code of the BatchNorm apply / backward
    passes; squeeze-excite = image sum + two tiny GEMMs + one scale pass (backward: one dot pass + one scale pass that also carries the gradient
    of the mean);
  * BatchNorm of the backbone: eps 1e-3, momentum 0.01 (torchvision's mobilenet_v3); of the head: torch's defaults;
  * recorded launch plans as in resnet_engine.py.
"""
from __future__ import annotations

import torch

from . import ops
from .deeplab_engine import DeepLabEngine
from .engine import _BN
from .ops import View
from .resnet_engine import Act, Unit

BIG = 1 << 20
BN_EPS, BN_MOMENTUM = 1e-3, 0.01
RE, HS = 1, ops.HARDSWISH
# (input, kernel, expanded, out, squeeze-excite, activation, stride, dilation): torchvision `_mobilenet_v3_conf("mobilenet_v3_large", dilated=True)`
BLOCKS = (
    (16, 3, 16, 16, False, RE, 1, 1), (16, 3, 64, 24, False, RE, 2, 1), (24, 3, 72, 24, False, RE, 1, 1), (24, 5, 72, 40, True, RE, 2, 1),
    (40, 5, 120, 40, True, RE, 1, 1), (40, 5, 120, 40, True, RE, 1, 1), (40, 3, 240, 80, False, HS, 2, 1), (80, 3, 200, 80, False, HS, 1, 1),
    (80, 3, 184, 80, False, HS, 1, 1), (80, 3, 184, 80, False, HS, 1, 1), (80, 3, 480, 112, True, HS, 1, 1), (112, 3, 672, 112, True, HS, 1, 1),
    (112, 5, 672, 160, True, HS, 2, 2), (160, 5, 960, 160, True, HS, 1, 2), (160, 5, 960, 160, True, HS, 1, 2),
)
LAST_CHANNELS = 960
MODE_OF_ACT = {0: 2, RE: 0, HS: ops.BN_MODE_HARDSWISH}      # activation behind a BatchNorm -> mode of its backward passes


def make_divisible(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def block_layers(i):
    """sub-module indices of block i inside torchvision's InvertedResidual.block: (expand | None, depthwise, squeeze-excite | None, project)"""
    cin, k, exp, cout, se, act, stride, dil = BLOCKS[i]
    j = 0
    e = None
    if exp != cin:
        e, j = j, j + 1
    d, j = j, j + 1
    s = None
    if se:
        s, j = j, j + 1
    return e, d, s, j


def _pad32(n):
    return -(-n // 32) * 32


class MobileNetEngine(DeepLabEngine):
    def __init__(self, num_classes: int, dtype: torch.dtype, device, aspp_dropout=0.0, dropout_seed=0):
        super().__init__("resnet18", num_classes, dtype, device, output_stride=16, plus=False, aspp_dropout=aspp_dropout, dropout_seed=dropout_seed)
        self.arch = "mobilenet_v3_large"
        self.rates = (12, 24, 36)          # torchvision's deeplabv3_mobilenet_v3_large: DeepLabHead's fixed rates at output stride 16
        self.overlap_wgrad = False

    # ------------------------------------------------------------------------------------------------ binding
    def bind(self, params, grads, buffers):
        """params / grads / buffers: the master tensors in torchvision's shapes.  self.P / self.G / self.Bf become what the launch plan reads:
        padded mirrors for the backbone, the masters themselves for the head."""
        self.MP, self.MG, self.MB = params, grads, buffers
        base = min(g.data_ptr() for g in grads.values())
        self._goff = {k: (g.data_ptr() - base) // 4 for k, g in grads.items()}
        dev, dt = self.dev, self.dtype
        kg = ops.KGROUP[ops.dtype_code(dt)]
        P, G, Bf = {}, {}, {}
        self.packed, self.bn = {}, {}
        entries, gath, scat, stat = [], [], [], []
        f32 = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)  # noqa: E731

        def vec(name, table_g, table_s, dstP, dstG=None):
            """a per-channel f32 vector: mirror padded to 32, gathered from the master; its gradient (dstG) scattered back"""
            src = (params if name in params else buffers)[name]
            n = src.numel()
            m = f32(_pad32(n))
            item = dict(src=src, dst=m, R=1, Cp=m.numel(), Rv=1, Cv=n, row=(1, BIG, 0, 0, 0), col=(1, BIG, 0, 0, 1), f32_out=True)
            table_g.append(item)
            dstP[name] = m
            if dstG is not None:
                g = f32(m.numel())
                table_s.append(dict(item, src=grads[name], dst=g))
                dstG[name] = g
            return m

        def gemm(key):
            """Conv2d [co, ci, k, k] as a GEMM over (ky, kx, ci) patches: forward operand [1][R][Kp], data-gradient operand [1][Kp][R], f32
            gradient [R][Kp]; R = co padded to 32, Kp = k*k*ci padded to 32"""
            w, g = params[key + ".weight"], grads[key + ".weight"]
            co, ci, k, _ = w.shape
            T = k * k
            R, K, Kp = _pad32(co), T * ci, _pad32(T * ci)
            wf = torch.zeros(1, R, Kp, dtype=dt, device=dev)
            wd = torch.zeros(1, Kp, max(R, kg), dtype=dt, device=dev)
            gw = f32(R, Kp)
            row, col = (1, BIG, 0, 0, ci * T), ((1, ci, 1, 0, T) if T > 1 else (1, BIG, 0, 0, 1))
            gath.append(dict(src=w, dst=wf, R=R, Cp=Kp, Rv=co, Cv=K, row=row, col=col))
            gath.append(dict(src=w, dst=wd, R=Kp, Cp=wd.shape[2], Rv=K, Cv=co, row=col, col=row))
            scat.append(dict(src=g, dst=gw, R=R, Cp=Kp, Rv=co, Cv=K, row=row, col=col, f32_out=True))
            self.packed[key] = dict(wf=wf, wd=wd)
            G[key + ".weight"] = gw.view(R, Kp, 1, 1)
            if (key + ".bias") in params:
                vec(key + ".bias", gath, scat, P, G)

        def depthwise(key):
            w, g = params[key + ".weight"], grads[key + ".weight"]
            c, _, k, _ = w.shape
            T, CP = k * k, _pad32(c)
            wt, gw = f32(T, CP), f32(T, CP)
            item = dict(src=w, dst=wt, R=T, Cp=CP, Rv=T, Cv=c, row=(1, BIG, 0, 0, 1), col=(1, BIG, 0, 0, T), f32_out=True)
            gath.append(item)
            scat.append(dict(item, src=g, dst=gw))
            self.packed[key] = dict(dw=wt)
            G[key + ".weight"] = gw

        def bn(key, backbone=True):
            if backbone:
                vec(key + ".weight", gath, scat, P, G)
                vec(key + ".bias", gath, scat, P, G)
                vec(key + ".running_mean", gath, stat, Bf, None)
                vec(key + ".running_var", gath, stat, Bf, None)
                for n_ in ("running_mean", "running_var"):      # mirror -> master after every train-mode forward
                    stat.append(dict(src=buffers[f"{key}.{n_}"], dst=Bf[f"{key}.{n_}"], R=1, Cp=Bf[f"{key}.{n_}"].numel(), Rv=1,
                                     Cv=buffers[f"{key}.{n_}"].numel(), row=(1, BIG, 0, 0, 0), col=(1, BIG, 0, 0, 1), f32_out=True))
            self.bn[key] = _BN(_pad32(params[key + ".weight"].numel()), dev)

        gemm("encoder.0.0"); bn("encoder.0.1")
        for i, (cin, k, exp, cout, se, act, stride, dil) in enumerate(BLOCKS):
            p = f"encoder.{i + 1}.block"
            e, d, s, pr = block_layers(i)
            if e is not None:
                gemm(f"{p}.{e}.0"); bn(f"{p}.{e}.1")
            depthwise(f"{p}.{d}.0"); bn(f"{p}.{d}.1")
            if s is not None:
                gemm(f"{p}.{s}.fc1"); gemm(f"{p}.{s}.fc2")
            gemm(f"{p}.{pr}.0"); bn(f"{p}.{pr}.1")
        last = f"encoder.{len(BLOCKS) + 1}"
        gemm(last + ".0"); bn(last + ".1")
        for name, w in params.items():
            if name.startswith("encoder."):
                continue
            P[name], G[name] = w, grads[name]
            if w.dim() == 4 and name != "classifier.weight":
                cout, cin, kh, kw = w.shape
                pk = dict(wf=torch.zeros(kh * kw, cout, max(cin, kg), dtype=dt, device=dev), wd=torch.zeros(kh * kw, cin, max(cout, kg), dtype=dt, device=dev))
                self.packed[name[:-len(".weight")]] = pk
                entries.append((w, pk["wf"], pk["wd"]))
            elif w.dim() == 1 and name.endswith(".weight"):
                bn(name[:-len(".weight")], backbone=False)
        for name, b in buffers.items():
            if not name.startswith("encoder."):
                Bf[name] = b
        self.P, self.G, self.Bf = P, G, Bf
        self._pack_table = ops.pack_table(entries, dev)
        self._gather = ops.gather_table(gath, dev)
        self._scatter = ops.gather_table(scat, dev)
        self._stat_scatter = ops.gather_table(stat, dev)
        self._keep = (gath, scat, stat)          # (the tables hold raw pointers into these tensors)
        cmax = max(max(m.numel() for m in P.values() if m.dim() == 1), 1024)
        self.one = torch.ones(cmax, dtype=torch.float32, device=dev)
        self.zero = torch.zeros(cmax, dtype=torch.float32, device=dev)
        self._rec, self._bufs, self.shape, self.gn = {}, {}, None, {}
        self._shape_cache.clear()

    def refresh_weights(self, train=True):
        ops.pack_conv_weights(self._pack_table[0], self._pack_table[1], self.dtype)
        ops.gather_weights(self._gather[0], self._gather[1], self.dtype)

    def enable_sync_bn(self, sync):
        self.sync_bn = sync
        cmax = max(p.numel() for p in self.P.values() if p.dim() == 1)
        self._sync_mom = torch.empty(3 * cmax, dtype=torch.float64, device=self.dev)
        self._sync_sums = torch.empty(2 * cmax, dtype=torch.float32, device=self.dev)
        self._rec = {}

    def _ready(self, name):
        """the backbone's gradients reach the master buffer in ONE scatter at the end of backward: no bucket of the flat gradient is complete
        before that (the data-parallel reducer flushes everything at optimiser time)"""

    def _bn_kw(self, bn):
        return dict(momentum=BN_MOMENTUM, eps=BN_EPS) if bn.startswith("encoder.") else {}

    # ------------------------------------------------------------------------------------------------ pieces
    def _unit(self, x: View, conv, bn, k, stride, pad, train, act_out: View | None, relu=True, virt=False, dil=1) -> Unit:
        """conv (GEMM | generic | depthwise, + batch statistics) -> BatchNorm finalize [-> apply + activation into act_out]; relu: 0 none,
        1 ReLU, 2 hardswish.  act_out None: the caller applies the BatchNorm (linear projection).  Eval mode: y / act_out hold bn(conv)."""
        act = int(relu)
        pk = self.packed[conv]
        dw = "dw" in pk
        B = x.B
        cout = x.C if dw else pk["wf"].shape[1]
        Ho, Wo = ops.conv_out_hw(x.H, x.W, k, k, stride, pad, dil)
        st, M, kw = self.bn[bn], B * Ho * Wo, self._bn_kw(bn)
        ops.SCOPE = "enc" if conv.startswith("encoder.") else "dec"
        gamma, beta, rm, rv = self.P[bn + ".weight"], self.P[bn + ".bias"], self.Bf[bn + ".running_mean"], self.Bf[bn + ".running_var"]
        y = ops.view(self._act(conv + ".y", B, Ho, Wo, cout))
        if not train:
            ops.bn_finalize(None, 0, M, cout, gamma, beta, rm, rv, False, st.scale, st.shift, None, None, **kw)
            out = act_out if act_out is not None else y
            if dw or act == HS:
                if dw:
                    ops.dwconv(x, pk["dw"], k, stride, pad, dil, y)
                else:
                    ops.conv2d(x, pk["wf"], None, y, k, k, stride, pad, dil)
                ops.bn_act(y, st.scale, st.shift, act, out)
            else:
                ops.conv2d(x, pk["wf"], None, out, k, k, stride, pad, dil, relu=act == RE and act_out is not None, pre_affine=(st.scale, st.shift))
            return Unit(x, out, conv, bn, k, stride, pad, False, dil)
        if dw:
            rows = ops.dwconv_rows(M, cout, self.dtype)
        else:
            rows = ops.conv_stat_rows(x, cout, k, k, stride, pad, dil)
        stats = (self._scratch("stat_sum", rows * cout), self._scratch("stat_m2", rows * cout), self._scratch("stat_cnt", rows))
        if dw:
            ops.dwconv(x, pk["dw"], k, stride, pad, dil, y, stats)
        else:
            ops.conv2d(x, pk["wf"], None, y, k, k, stride, pad, dil, stats=stats)
        need = ops.bn_finalize_workspace_floats(rows, cout)
        if self.sync_bn is None:
            ops.bn_finalize(stats, rows, M, cout, gamma, beta, rm, rv, True, st.scale, st.shift, st.mean, st.invstd,
                            workspace=self._scratch("bn_ws", max(need, 4)), **kw)
        else:
            mom = self._sync_mom[:3 * cout]
            ops.bn_moments(stats, rows, cout, mom, workspace=self._scratch("bn_ws", max(need, 4)))
            self._host(lambda mom=mom: self.sync_bn.all_reduce(mom))
            ops.bn_finalize_moments(mom, cout, gamma, beta, rm, rv, st.scale, st.shift, st.mean, st.invstd, **kw)
        if act_out is not None:
            ops.bn_act(y, st.scale, st.shift, act, act_out)
            if act == RE:
                self.relu_order.append(act_out)      # every ReLU output in execution order (tests evaluate their float64 reference at these decisions)
            elif act == HS:
                self.hs_order.append(act_out)        # ... and every hardswish output (its derivative jumps at -3 and 3)
        u = Unit(x, y, conv, bn, k, stride, pad, False, dil)
        u.dw, u.act, u.fp8 = dw, act, False
        self.units[conv] = (u, act_out)
        return u

    def _wgrad(self, u: Unit, dy: View, ws_name):
        if getattr(u, "dw", False):
            T = u.k * u.k
            rows = ops.dwconv_wgrad_rows(dy.B * dy.H * dy.W, dy.C, u.k, self.dtype)
            part = self._scratch("dw_part", rows * T * dy.C)
            ops.dwconv_wgrad(u.x, dy, u.k, u.stride, u.pad, u.dil, part)
            ops.colsum_finalize(part, rows, T * dy.C, self.G[u.conv + ".weight"])
        else:
            super()._wgrad(u, dy, ws_name)

    def _dgrad(self, u: Unit, dy: View, name, fuse_into: Unit | None = None):
        if getattr(u, "dw", False):
            x = u.x
            gx = ops.view(self._act(name, x.B, x.H, x.W, x.C))
            ops.SCOPE = "enc"
            ops.dwconv_dgrad(dy, self.packed[u.conv]["dw"], u.k, u.stride, u.pad, u.dil, gx)
            return gx, False
        return super()._dgrad(u, dy, name, fuse_into)

    def _fusable(self, producer, consumer):
        return False

    def _fc(self, x: View, key, out: View, relu=False):
        ops.SCOPE = "enc"
        ops.conv2d(x, self.packed[key]["wf"], self.P[key + ".bias"], out, 1, 1, relu=relu)

    def _fc_bwd(self, x: View, g: View, key, gx_name):
        """weight + bias gradient of a squeeze-excite FC ([B,1,1,C] maps); returns the data gradient"""
        ops.SCOPE = "enc"
        gw, gb = self.G[key + ".weight"], self.G[key + ".bias"]
        fused = ops.wgrad_takes_bias(x, g, 1, 1, 1, 0)
        need = ops.wgrad_workspace_floats_for(x, g, 1, 1, 1, 0, with_bias=fused)
        ops.conv2d_wgrad(x, g, gw, 1, 1, 1, 0, self._scratch("wg_ws", need), dbias=gb if fused else None)
        if not fused:
            rows = ops.bn_bwd_rows(g.B)
            part = self._scratch("bias_part", rows * g.C)
            ops.colsum_partial(g, part)
            ops.colsum_finalize(part, rows, g.C, gb)
        gx = ops.view(self._act(gx_name, x.B, 1, 1, x.C))
        ops.conv2d(g, self.packed[key]["wd"], None, gx, 1, 1)
        return gx

    def _se(self, a: Act, p):
        """torchvision SqueezeExcitation on the activated depthwise output: -> (scaled activation, backward closure(g) -> gradient of a)"""
        v = a.v
        B, H, W, C_ = v.B, v.H, v.W, v.C
        sq = self.packed[p + ".fc1"]["wf"].shape[1]
        pooled = ops.view(self._act(p + ".pool", B, 1, 1, C_))
        ops.image_sum(v, pooled, 1.0 / (H * W))
        r = ops.view(self._act(p + ".r", B, 1, 1, sq))
        self._fc(pooled, p + ".fc1", r, relu=True)
        self.relu_order.append(r)
        t = ops.view(self._act(p + ".t", B, 1, 1, C_))
        self._fc(r, p + ".fc2", t)
        s = ops.view(self._act(p + ".s", B, 1, 1, C_))
        ops.hardsigmoid(t, s)
        out = ops.view(self._act(p + ".out", B, H, W, C_))
        ops.se_scale(v, s, out)

        def bwd(g: View) -> View:
            gs = ops.view(self._act(p + ".gs", B, 1, 1, C_))
            ops.image_dot(g, v, gs)
            gt = ops.view(self._act(p + ".gt", B, 1, 1, C_))
            ops.hardsigmoid(t, gt, g=gs)
            gr_raw = self._fc_bwd(r, gt, p + ".fc2", p + ".gr_raw")
            gr = ops.view(self._act(p + ".gr", B, 1, 1, sq))
            ops.relu_bwd_sum(r, [(gr_raw, False)], gr)
            gpool = self._fc_bwd(pooled, gr, p + ".fc1", p + ".gpool")
            ga = ops.view(self._act(p + ".ga", B, H, W, C_))
            ops.se_scale(g, s, ga, add=gpool, add_scale=1.0 / (H * W))
            return ga
        return out, bwd

    def _sum_grads(self, a: Act, name) -> View:
        assert 1 <= len(a.grads) <= 3 and not any(half for _, half in a.grads)
        if len(a.grads) == 1:
            return a.grads[0][0]
        v = a.v
        g = ops.view(self._act(name, v.B, v.H, v.W, v.C))
        ops.relu_bwd_sum(None, a.grads, g)
        return g

    def _block(self, h: Act, i, train, tape) -> Act:
        cin, k, exp, cout, se, act, stride, dil = BLOCKS[i]
        e, d, s, pr = block_layers(i)
        pre = f"encoder.{i + 1}.block"
        B, CE = h.v.B, _pad32(exp)
        t, ue = h, None
        if e is not None:
            t = Act(ops.view(self._act(f"{pre}.{e}.a", B, h.v.H, h.v.W, CE)))
            ue = self._unit(h.v, f"{pre}.{e}.0", f"{pre}.{e}.1", 1, 1, 0, train, t.v, relu=act)
        st, pad = (1 if dil > 1 else stride), (k - 1) // 2 * dil          # (torchvision: a dilated block keeps stride 1)
        Ho, Wo = ops.conv_out_hw(h.v.H, h.v.W, k, k, st, pad, dil)
        ad = Act(ops.view(self._act(f"{pre}.{d}.a", B, Ho, Wo, CE)))
        ud = self._unit(t.v, f"{pre}.{d}.0", f"{pre}.{d}.1", k, st, pad, train, ad.v, relu=act, dil=dil)
        z, se_bwd = ad.v, None
        if s is not None:
            z, se_bwd = self._se(ad, f"{pre}.{s}")
        up = self._unit(z, f"{pre}.{pr}.0", f"{pre}.{pr}.1", 1, 1, 0, train, None, relu=0)
        res = stride == 1 and cin == cout
        out = Act(ops.view(self._act(f"{pre}.out", B, Ho, Wo, up.y.C)))
        sp = self.bn[up.bn]
        if train:
            ops.bn_add(up.y, sp.scale, sp.shift, h.v if res else None, out.v)
        elif res:
            ops.bn_add(up.y, self.one[:up.y.C], self.zero[:up.y.C], h.v, out.v)
        else:
            out = Act(up.y)
        if train:
            def bwd():
                g = self._sum_grads(out, f"{pre}.gsum")
                dy = self._unit_bwd(up, g, MODE_OF_ACT[0])
                gz, _ = self._dgrad(up, dy, f"{pre}.g_z")
                if se_bwd is not None:
                    gz = se_bwd(gz)
                dy = self._unit_bwd(ud, gz, MODE_OF_ACT[act])
                gt, _ = self._dgrad(ud, dy, f"{pre}.g_t")
                if ue is not None:
                    dy = self._unit_bwd(ue, gt, MODE_OF_ACT[act])
                    gt, _ = self._dgrad(ue, dy, f"{pre}.g_h")
                h.grads.append((gt, False))
                if res:
                    h.grads.append((g, False))
            tape.append(bwd)
        return out

    def _encoder(self, B, S, train, tape):
        Kp = self.packed["encoder.0.0"]["wf"].shape[2]
        col = ops.view(self._buf("stem_col", (B, S // 2, S // 2, Kp)))          # written by the eager cvcs_im2col_stem launch of _forward_backbone
        h = Act(ops.view(self._act("encoder.0.a", B, S // 2, S // 2, self.packed["encoder.0.0"]["wf"].shape[1])))
        u0 = self._unit(col, "encoder.0.0", "encoder.0.1", 1, 1, 0, train, h.v, relu=HS)
        if train:
            def stem_bwd(h=h):      # (bound now: `h` is rebound by the block loop below)
                g = self._sum_grads(h, "encoder.0.gsum")
                self._unit_bwd(u0, g, MODE_OF_ACT[HS])
                # every gradient of the backbone is complete: one launch moves them into the master layout
                ops.scatter_weight_grads(self._scatter[0], self._scatter[1])
            tape.append(stem_bwd)
        for i in range(len(BLOCKS)):
            h = self._block(h, i, train, tape)
        last = f"encoder.{len(BLOCKS) + 1}"
        f5 = Act(ops.view(self._act(last + ".a", B, h.v.H, h.v.W, LAST_CHANNELS)))
        ul = self._unit(h.v, last + ".0", last + ".1", 1, 1, 0, train, f5.v, relu=HS)
        if train:
            ops.scatter_weight_grads(self._stat_scatter[0], self._stat_scatter[1])      # running statistics: mirrors -> masters
            prev = h

            def last_bwd():
                g = self._sum_grads(f5, last + ".gsum")
                dy = self._unit_bwd(ul, g, MODE_OF_ACT[HS])
                prev.grads.append(self._dgrad(ul, dy, last + ".g_h"))
            tape.append(last_bwd)
        return None, f5

    def _backbone(self, B, S, train):
        self._q8, self._train_pass = {}, train
        self.hs_order = []
        return super()._backbone(B, S, train)

    def _forward_backbone(self, x, train):
        B, C_, S, S2 = x.shape
        assert C_ == 3 and S == S2
        self._plan(B, S)
        Kp = self.packed["encoder.0.0"]["wf"].shape[2]
        ops.im2col_stem(x.contiguous(), ops.view(self._buf("stem_col", (B, S // 2, S // 2, Kp))), 3, 3, 2, 1, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0))
        key = "fwd_train" if train else "fwd_eval"
        if train and "fwd_train" not in self._rec:
            self._rec.pop("bwd", None)
        self._run(key, lambda: self._backbone(B, S, train))
        if key not in self._last_act:
            self._last_act[key] = self._last
        self._saved_train = train
        return self._last_act[key]
