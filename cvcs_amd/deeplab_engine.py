"""Launch plan of the DeepLab family on the HIP kernels: DeepLabV3+ (BASELINE.json configs[2]: "DeepLabV3+ (ResNet50, ASPP
dilations 6/12/18)") and the DeepLabV3 the reference wraps from torchvision under its factory name `Resnet101`
(S/nets.py:234-257, S/utils.py:180-181; the wrapper returns d['out'] :248-250).

Everything up to the last encoder stage is the ResNet plan of resnet_engine.py (recorded launch lists, NHWC bf16/f32, fused
block tails), with torchvision's stride -> dilation replacement in the last stage(s).  On top of it:
  * ASPP: the five branches write into the channel ranges of ONE [B,h,w,1280] buffer (no concat pass): 1x1 branch on the
    non-overlapping-taps kernel, the three atrous 3x3 branches on the generic gather kernel (at rates 6..36 the nine taps of
    a 16x16 tile touch nine disjoint windows - there is no halo to share, see wgrad_dilated in wgrad.hip), the image-pooling
    branch as per-image mean -> 1x1 conv -> BN -> ReLU -> broadcast; then the 1x1 projection;
  * V3+ decoder: 1x1 reduction of the stride-4 feature into the second channel range of the decoder's input buffer, the
    ASPP output resized x4 into the first, two 3x3 convs on the halo kernel, 1x1 classifier at stride 4 (NCHW f32 logits),
    bilinear x4 to the input size; backward mirrors it with gather-form resize gradients;
  * V3 head (plus=False): ASPP -> 3x3 conv -> classifier -> bilinear x output_stride.
"""
from __future__ import annotations

import torch

from . import ops
from .ops import View
from .resnet_engine import Act, ResNetUNetEngine

LOW_LEVEL_CHANNELS = 64


class DeepLabEngine(ResNetUNetEngine):
    def __init__(self, arch, num_classes, dtype, device, output_stride=16, plus=True, aspp_dropout=0.0, dropout_seed=0):
        super().__init__(arch, num_classes, dtype, device)
        assert output_stride in (8, 16) and 0.0 <= aspp_dropout < 1.0
        # nn.Dropout(p) behind the ASPP projection in train mode (torchvision's ASPP has p = 0.5: the reference's `Resnet101` trains with it,
        # S/nets.py:242-246): a counter-based mask, state = [seed, step] on the device (ops.dropout); the step advances once per backward
        self.aspp_dropout = float(aspp_dropout)
        self.drop_state = torch.tensor([dropout_seed, 0], dtype=torch.int64, device=self.dev)
        self.os, self.plus = output_stride, plus
        self.rates = (6, 12, 18) if output_stride == 16 else (12, 24, 36)
        self.head_name = "classifier"

    # ------------------------------------------------------------------------------------------------ the network
    def _backbone(self, B, S, train):
        tape = []
        self.units, self.relu_order = {}, []
        self.refresh_weights()
        OS = self.os
        h5 = S // OS
        f2, f5 = self._encoder(B, S, train, tape)
        C5 = f5.v.C
        # ---- ASPP: five branches into one buffer
        cat = self._act("aspp.cat", B, h5, h5, 5 * 256)
        br = [Act(View(cat, 256 * i, 256)) for i in range(5)]
        units = [self._unit(f5.v, "aspp.convs.0.0", "aspp.convs.0.1", 1, 1, 0, train, br[0].v)]
        for i, r in enumerate(self.rates, start=1):
            units.append(self._unit(f5.v, f"aspp.convs.{i}.0", f"aspp.convs.{i}.1", 3, 1, r, train, br[i].v, dil=r))
        pooled = Act(ops.view(self._act("aspp.pooled", B, 1, 1, C5)))
        ops.image_sum(f5.v, pooled.v, 1.0 / (h5 * h5))
        pa = Act(ops.view(self._act("aspp.pool_act", B, 1, 1, 256)))
        up = self._unit(pooled.v, "aspp.convs.4.1", "aspp.convs.4.2", 1, 1, 0, train, pa.v)
        ops.image_broadcast(pa.v, br[4].v, 1.0)
        a = Act(ops.view(self._act("aspp.out", B, h5, h5, 256)))
        uproj = self._unit(ops.view(cat), "aspp.project.0", "aspp.project.1", 1, 1, 0, train, a.v)
        drop = train and self.aspp_dropout > 0.0
        if drop:      # in place: the projection's own backward reads its conv output, not `a`; every consumer of `a` sees the dropped tensor
            ops.dropout(a.v, a.v, self.drop_state, self.aspp_dropout)
        if train:
            def aspp_bwd():
                assert len(a.grads) == 1
                if drop:
                    ops.dropout(a.grads[0][0], a.grads[0][0], self.drop_state, self.aspp_dropout)   # the same mask on the gradient
                dy = self._unit_bwd(uproj, a.grads[0][0], 0)
                gcat, _ = self._dgrad(uproj, dy, "aspp.g_cat")
                # image-pooling branch: broadcast^T = per-image sum; conv + BN backward on [B,1,1,256]; mean^T = broadcast / HW
                gpa = ops.view(self._act("aspp.g_pool_act", B, 1, 1, 256))
                ops.image_sum(View(gcat.t, 4 * 256, 256), gpa, 1.0)
                dyp = self._unit_bwd(up, gpa, 0)
                gpooled, _ = self._dgrad(up, dyp, "aspp.g_pooled")
                g5p = ops.view(self._act("aspp.g_f5_pool", B, h5, h5, C5))
                ops.image_broadcast(gpooled, g5p, 1.0 / (h5 * h5))
                grads = [(g5p, False)]
                for i in range(4):
                    dyi = self._unit_bwd(units[i], View(gcat.t, 256 * i, 256), 0)
                    grads.append(self._dgrad(units[i], dyi, f"aspp.g_f5_{i}"))
                # f5 has five gradient sources: summed three at a time (cvcs_relu_bwd_sum without a mask, then with f5's own)
                part = ops.view(self._act("aspp.g_f5_sum", B, h5, h5, C5))
                ops.relu_bwd_sum(None, grads[:3], part)
                f5.grads.append((part, False))
                f5.grads.extend(grads[3:])
                self._ready("aspp.convs.0.0.weight")
            tape.append(aspp_bwd)
        if self.plus:
            s4, up_f = S // 4, OS // 4
            dcat = self._act("dec.cat", B, s4, s4, 256 + LOW_LEVEL_CHANNELS)
            low = Act(View(dcat, 256, LOW_LEVEL_CHANNELS))
            ulow = self._unit(f2.v, "low_level.0", "low_level.1", 1, 1, 0, train, low.v)
            ops.resize_bilinear_fwd(a.v, View(dcat, 0, 256), up_f)
            d1 = Act(ops.view(self._act("dec.a1", B, s4, s4, 256)))
            d2 = Act(ops.view(self._act("dec.a2", B, s4, s4, 256)))
            u1 = self._unit(ops.view(dcat), "decoder.conv1.0", "decoder.conv1.1", 3, 1, 1, train, d1.v)
            u2 = self._unit(d1.v, "decoder.conv2.0", "decoder.conv2.1", 3, 1, 1, train, d2.v)
            if train:
                def dec_bwd():
                    assert len(d2.grads) == 1
                    dy = self._unit_bwd(u2, d2.grads[0][0], 0)
                    g1, _ = self._dgrad(u2, dy, "dec.g_a1")
                    dy = self._unit_bwd(u1, g1, 0)
                    gcat, _ = self._dgrad(u1, dy, "dec.g_cat")
                    ga = ops.view(self._act("dec.g_aspp", B, h5, h5, 256))
                    ops.resize_bilinear_bwd(View(gcat.t, 0, 256), ga, up_f)
                    a.grads.append((ga, False))
                    dyl = self._unit_bwd(ulow, View(gcat.t, 256, LOW_LEVEL_CHANNELS), 0)
                    f2.grads.append(self._dgrad(ulow, dyl, "dec.g_f2"))
                    self._ready("low_level.0.weight")
                tape.append(dec_bwd)
            last, self._final = d2, 4
        else:
            d1 = Act(ops.view(self._act("head.a1", B, h5, h5, 256)))
            u1 = self._unit(a.v, "head.0", "head.1", 3, 1, 1, train, d1.v)
            if train:
                def head_bwd():
                    assert len(d1.grads) == 1
                    dy = self._unit_bwd(u1, d1.grads[0][0], 0)
                    a.grads.append(self._dgrad(u1, dy, "head.g_aspp"))
                    self._ready("head.0.weight")
                tape.append(head_bwd)
            last, self._final = d1, OS
        if train:
            self._tape = tape
        self._last = last
        return last

    def _encoder(self, B, S, train, tape):
        """-> (stride-4 feature, output-stride feature): the dilated ResNet (the MobileNetV3 engine overrides this)"""
        f1, p0 = self._stem(B, S, train, tape)
        feats = self._stages(p0, train, tape, lambda s: None, output_stride=self.os)
        return feats[1], feats[4]

    def _plan(self, B, S):
        if self.shape != (B, S):
            assert S % 32 == 0, "tile side must be a multiple of 32"
            known = (B, S) in self._shape_cache
            super()._plan(B, S)
            if not known:
                f = 4 if self.plus else self.os
                self.logits_lo = torch.empty((B, self.NC, S // f, S // f), dtype=torch.float32, device=self.dev)
                self.dlogits_lo = torch.empty_like(self.logits_lo)

    def forward(self, x: torch.Tensor, train: bool) -> torch.Tensor:
        if train and x.shape[0] < 2:   # (torchvision's ASPPPooling BatchNorm raises the same on a 1x1 map of one image)
            raise ValueError("Expected more than 1 value per channel when training: DeepLab needs a batch of at least 2 tiles")
        h = self._forward_backbone(x, train)
        w = self.P["classifier.weight"].view(self.NC, 256)
        ops.linear_head_fwd(h.v, w, self.P["classifier.bias"], logits=self.logits_lo)
        ops.resize_bilinear_nchw_fwd(self.logits_lo, self.logits, 4 if self.plus else self.os)
        return self.logits

    def forward_labels(self, x: torch.Tensor, labels: torch.Tensor):
        """argmax of the RESIZED logits (the resize mixes neighbouring pixels' logits: it cannot be fused away)"""
        logits = self.forward(x, False)
        ops.argmax_confusion(logits, labels)
        return labels

    def backward(self, dlogits: torch.Tensor):
        assert self._saved_train, "backward needs a preceding forward in train mode"
        if self.on_backward_begin is not None:
            self.on_backward_begin()
        B, S = self.shape
        NC, f = self.NC, (4 if self.plus else self.os)
        h = self._last_act["fwd_train"]
        ops.resize_bilinear_nchw_bwd(dlogits.contiguous(), self.dlogits_lo, f)
        P = B * (S // f) * (S // f)
        rows = ops.linear_head_bwd_rows(P)
        part = self._scratch("head_part", rows * (NC * 256 + NC))
        gh = ops.view(self._act("head.gx", B, S // f, S // f, 256))
        w = self.P["classifier.weight"].view(NC, 256)
        ops.linear_head_bwd(h.v, self.dlogits_lo, w, gh, part)
        gw, gb = self.G["classifier.weight"], self.G["classifier.bias"]
        assert gb.data_ptr() == gw.data_ptr() + gw.numel() * 4
        ops.colsum_finalize(part, rows, NC * 256 + NC, gw)
        if self.on_grad_ready is not None:
            self.on_grad_ready(self._goff["classifier.weight"], [])

        def run_tape():
            h.grads = [(gh, False)]
            for fn in reversed(self._tape):
                fn()
            if self.aspp_dropout > 0.0:
                ops.counter_add(self.drop_state, 1)      # the next step draws a new mask
        self._run("bwd", run_tape)
