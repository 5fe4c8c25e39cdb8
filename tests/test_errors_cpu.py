"""CPU-only: error behaviour of the boundary.  Every C-ABI entry point validates its arguments on the host before any HIP
call (so these run without a GPU) and reports through the int status + cvcs_last_error(); the Python factories keep the
reference's error behaviour (S/utils.py:174-221)."""
import ctypes as C
import os

import pytest
import torch

from cvcs_amd import _lib, nets, utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.lib()


def _buf(n=4096):
    b = (C.c_char * (n + 64))()
    a = (C.addressof(b) + 63) // 64 * 64
    return b, a           # keep b alive; a is 64-byte aligned


def _desc(a, **kw):
    d = _lib.ConvDesc()
    d.in_, d.in_ld, d.B, d.H, d.W, d.Cin = a, 32, 1, 8, 8, 32
    d.wt, d.bias, d.out, d.out_ld, d.Ho, d.Wo, d.Cout = a, None, a, 64, 8, 8, 64
    d.KH, d.KW, d.stride, d.pad, d.dil, d.relu, d.pixel_shuffle, d.dtype = 3, 3, 1, 1, 1, 0, 0, 1   # 1 = bf16
    for k, v in kw.items():
        setattr(d, k, v)
    return d


@pytest.mark.parametrize("kw,msg", [
    (dict(dtype=7), b"bad dtype"),
    (dict(Cin=24, in_ld=24), b"Cin=24 must be a multiple of 32"),
    (dict(Cout=40), b"Cout=40 must be a multiple of 32"),      # (16 | 32 | 48 run as masked thin tiles of the 3x3 halo kernel)
    (dict(Ho=7), b"geometry gives 8,8"),
    (dict(B=0), b"bad shape"),
    (dict(in_ld=16), b"in_ld"),
    (dict(out_ld=60), b"out_ld"),
    (dict(stride=0), b"bad filter geometry"),
    (dict(pixel_shuffle=1, Cout=128), b"pixel_shuffle needs"),
])
def test_conv2d_rejects_bad_descriptors(lib, kw, msg):
    keep, a = _buf()
    rc = lib.cvcs_conv2d(C.byref(_desc(a, **kw)), None)
    assert rc == -1 and msg in lib.cvcs_last_error(), lib.cvcs_last_error()


def test_conv2d_rejects_misaligned_and_half_given_arguments(lib):
    keep, a = _buf()
    assert lib.cvcs_conv2d(C.byref(_desc(a, in_=a + 2)), None) == -1 and b"16-byte aligned" in lib.cvcs_last_error()
    assert lib.cvcs_conv2d(C.byref(_desc(a, stat_sum=a)), None) == -1 and b"go together" in lib.cvcs_last_error()
    assert lib.cvcs_conv2d(C.byref(_desc(a, pre_scale=a)), None) == -1 and b"both scale and shift" in lib.cvcs_last_error()
    assert lib.cvcs_conv2d(C.byref(_desc(a, wt=None)), None) == -1 and b"null tensor" in lib.cvcs_last_error()


def test_wgrad_rejects_unbuilt_filters_and_bad_shapes(lib):
    keep, a = _buf()
    d = _lib.WgradDesc()
    d.x, d.x_ld, d.B, d.H, d.W, d.Cin = a, 64, 1, 8, 8, 64
    d.dy, d.dy_ld, d.Ho, d.Wo, d.Cout = a, 64, 8, 8, 64
    d.KH, d.KW, d.stride, d.pad, d.dw, d.Cin_real, d.workspace, d.dtype = 5, 5, 1, 2, a, 64, a, 1
    assert lib.cvcs_conv2d_wgrad(C.byref(d), None) == -1 and b"filter 5x5 not built" in lib.cvcs_last_error()
    d.KH = d.KW = 3
    d.pad = 0
    assert lib.cvcs_conv2d_wgrad(C.byref(d), None) == -1 and b"geometry gives 6,6" in lib.cvcs_last_error()
    d.pad, d.Cout = 1, 40
    assert lib.cvcs_conv2d_wgrad(C.byref(d), None) == -1 and b"Cout=40" in lib.cvcs_last_error()
    assert lib.cvcs_wgrad_slices(0, 8, 8, 64, 64, 3, 3, 1) == -1


def test_other_entry_points_reject_bad_arguments(lib):
    keep, a = _buf()
    assert lib.cvcs_head_fwd(a, 48, 1, 4, 4, 48, a, a, 5, a, 1, None) == -1 and b"input channels, got 48" in lib.cvcs_last_error()
    assert lib.cvcs_head_argmax(a, 64, 1, 4, 4, 64, a, a, 40, a, 1, None) == -1 and b"NC=40" in lib.cvcs_last_error()
    assert lib.cvcs_ce_fwd_bwd(a, a, 1, 1, 33, 16, None, 0, 1.0, a, a, a, 0, None) == -1 and b"NC=33" in lib.cvcs_last_error()
    assert lib.cvcs_vote_labels(a, 0, 16, a, None) == -1 and b"voters" in lib.cvcs_last_error()
    assert lib.cvcs_vote_labels(a, 17, 16, a, None) == -1
    assert lib.cvcs_label_confusion(a, a, 1, 16, 0, 64, a, None) == -1 and b"K=64" in lib.cvcs_last_error()
    assert lib.cvcs_bn_act(a, 64, 1, 3, 4, 64, a, a, 1, a, 64, a, 64, 1, None) == -1 and b"even H, W" in lib.cvcs_last_error()
    assert lib.cvcs_bn_act(a, 64, 1, 4, 4, 64, a, a, 1, None, 0, None, 0, 1, None) == -1 and b"nothing to write" in lib.cvcs_last_error()
    assert lib.cvcs_crop_tiles(a, 3, 8, 8, a, 1, 0, 1, 4, 5, None) == -1          # S - p must be even
    assert lib.cvcs_label_stitch(a, 1, 8, 4, 0, 1, None, a, None, 8, 8, None) == -1 and b"palette" in lib.cvcs_last_error()
    assert lib.cvcs_bn_finalize(None, None, None, 0, 0, 60, a, a, a, a, 0.1, 1e-5, 0, a, a, None, None, None, None) == -1


def test_mobilenet_entry_points_reject_bad_arguments(lib):
    """depthwise / squeeze-excite entry points (csrc/mobilenet.hip): kernel sizes other than 3 / 5, an output size that is not the
    convolution's, half-given statistics pointers, misaligned views; the row queries the engine sizes its scratch buffers with"""
    keep, a = _buf(1 << 16)
    # x [1,8,8,32] bf16, 3x3 / stride 1 / pad 1 -> y [1,8,8,32]
    ok = (a, 32, 1, 8, 8, 32, a, 3, 1, 1, 1, a, 32, 8, 8, None, None, None, 1, None)
    bad_k = ok[:7] + (4,) + ok[8:]
    assert lib.cvcs_dwconv(*bad_k) == -1 and b"kernel size 3 or 5" in lib.cvcs_last_error()
    bad_out = ok[:13] + (7, 8) + ok[15:]
    assert lib.cvcs_dwconv(*bad_out) == -1 and b"output size mismatch" in lib.cvcs_last_error()
    half_stats = ok[:15] + (a, None, None) + ok[18:]
    assert lib.cvcs_dwconv(*half_stats) == -1 and b"statistics pointers" in lib.cvcs_last_error()
    misaligned = (a + 2,) + ok[1:]
    assert lib.cvcs_dwconv(*misaligned) == -1 and b"16-byte aligned" in lib.cvcs_last_error()
    assert lib.cvcs_dwconv_dgrad(a, 32, 1, 4, 4, 32, a, 3, 2, 1, 1, a, 32, 9, 8, 1, None) == -1 and b"output size mismatch" in lib.cvcs_last_error()
    assert lib.cvcs_dwconv_wgrad(a, 32, a, 32, 1, 8, 8, 32, 7, 1, 3, 1, 8, 8, a, 1, None) == -1 and b"kernel size 3 or 5" in lib.cvcs_last_error()
    assert lib.cvcs_dwconv_wgrad(a, 32, a, 32, 1, 8, 8, 36, 3, 1, 1, 1, 8, 8, a, 1, None) == -1 and b"bad shape" in lib.cvcs_last_error()      # C % 8 (bf16)
    assert lib.cvcs_se_scale(a, 32, None, 32, None, 0, 0.0, 1, 64, 32, a, 32, 1, None) == -1 and b"null tensor" in lib.cvcs_last_error()
    assert lib.cvcs_image_dot(a, 16, a, 32, 1, 64, 32, a, 32, 1, None) == -1 and b"16-byte aligned" in lib.cvcs_last_error()                 # ld < C
    assert lib.cvcs_hardsigmoid(a, 32, None, 0, 0, 32, a, 32, 1, None) == -1 and b"bad shape" in lib.cvcs_last_error()
    assert lib.cvcs_bn_add(a, 32, None, a, None, 0, 64, 32, a, 32, 1, None) == -1 and b"bad shape" in lib.cvcs_last_error()
    assert lib.cvcs_bn_bwd_reduce(a, 32, a, 32, None, 0, 1, 8, 8, 32, a, a, a, a, 4, a, a, 1, None) == -1 and b"bad mode" in lib.cvcs_last_error()  # modes 0..3
    # row queries: at least one row, never more workgroups than the caps (8192 forward / 4096 weight gradient over rows x slabs [x K])
    assert lib.cvcs_dwconv_rows(0, 32, 1) == 0 and lib.cvcs_dwconv_rows(64, 36, 1) == 0 and lib.cvcs_dwconv_wgrad_rows(64, 32, 4, 1) == 0
    for M, C_, K in ((1, 32, 3), (64, 960, 5), (32 * 256 * 256, 32, 3), (32 * 64 * 64, 96, 5), (32 * 32 * 32, 672, 5)):
        for dt, V in ((0, 4), (1, 8)):
            r, rw = lib.cvcs_dwconv_rows(M, C_, dt), lib.cvcs_dwconv_wgrad_rows(M, C_, K, dt)
            cc = C_ // V
            lanes = max(d for d in range(1, 33) if cc % d == 0)
            assert 1 <= r <= max(64, 8192 // (cc // lanes)) and 1 <= rw <= max(64, 4096 // ((cc // lanes) * K)), (M, C_, K, dt, r, rw)


def test_factories_keep_the_reference_error_behaviour(capsys):
    with pytest.raises(Exception):
        utils.load_network({"net": "NoSuchNet", "num_classes": 4}, "cpu")
    assert "Invalid network name." in capsys.readouterr().out                       # S/utils.py:193-195
    mb = utils.load_network({"net": "MobileNet", "num_classes": 4}, "cpu")           # S/utils.py:183-184: every factory name of the reference is built
    assert isinstance(mb, nets.DeepLabV3MobileNet) and mb.wrapper and mb.num_classes == 5
    with pytest.raises(Exception):
        utils.load_network({"net": "Ensemble", "num_classes": 4}, "cpu")            # no ensemble_config (S/utils.py:475-478)
    assert "specify a config file" in capsys.readouterr().out
    net = nets.Urnetv2(5)
    with pytest.raises(ValueError, match="Optimizer name not valid"):
        utils.load_optimizer({"opt": "LION", "epochs": 1}, net)                     # S/utils.py:219-220
    with pytest.raises(Exception):
        utils.load_loss({"loss": "Dice", "num_classes": 4}, "cpu")
    with pytest.raises(AssertionError):
        nets.DeepLabV3Plus(5, precision="fp8")                                      # fp8 is built for the U-Nets, ResNet-UNets, Swin + UPerNet
    assert nets.Urnetv2(5, precision="fp8").precision == "fp8"
    with pytest.raises(RuntimeError, match="GPU only"):                             # no CPU fallback
        net(torch.zeros(1, 3, 32, 32))
    assert utils.load_device({"device": "cpu"}) == torch.device("cpu")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """no CPU fallback: without libcvcs_hip.so the first op raises CvcsError naming the build command (tier rule 3)"""
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libcvcs_hip.so"))
    with pytest.raises(_lib.CvcsError, match="There is no CPU fallback"):
        _lib.lib()
    from cvcs_amd import ops
    with pytest.raises(_lib.CvcsError):
        ops.bn_bwd_rows(10)


def test_check_turns_a_status_into_an_exception(lib):
    assert lib.cvcs_conv2d(None, None) == -1
    with pytest.raises(_lib.CvcsError, match="null descriptor"):
        _lib.check(-1, "cvcs_conv2d")


def test_batchnorm_backward_lane_rule_covers_every_channel_count_exactly(lib):
    """the geometry behind round 3's memory-access fault (MobileNet bring-up, cvcs_bn_bwd_reduce on 4 x 4-pixel f32 maps of 96 channels): the
    channel-chunk lanes of a workgroup must DIVIDE the row's 16-byte chunk count, so that gridDim.y * lanes covers a pixel row exactly and no
    thread is handed a chunk (or, through the left-over pixel lane, a pixel) beyond the tensor - for every width, not only powers of two"""
    for dtype, V in ((_lib.F32, 4), (_lib.BF16, 8)):
        for C_ in range(V, 2049, V):
            lanes = lib.cvcs_bn_bwd_chunk_lanes(C_, dtype)
            CC = C_ // V
            assert 1 <= lanes <= 32 and CC % lanes == 0, (C_, dtype, lanes)
            assert lanes * (256 // lanes) <= 256          # the pixel lanes fit the workgroup; the remainder idles (kernel: `pl < PL`)
        assert lib.cvcs_bn_bwd_chunk_lanes(V + 1, dtype) == -1 and lib.cvcs_bn_bwd_chunk_lanes(0, dtype) == -1
    assert lib.cvcs_bn_bwd_chunk_lanes(96, _lib.F32) == 24 and lib.cvcs_bn_bwd_chunk_lanes(960, _lib.F32) == 30


def test_round4_entry_points_reject_bad_arguments(lib):
    keep, a = _buf(1 << 16)
    # residual / second-source / mask epilogues are bf16 1x1 features of cvcs_conv2d
    d = _desc(a, res=a, res_ld=64)
    assert lib.cvcs_conv2d(C.byref(d), None) == -1 and b"1x1" in lib.cvcs_last_error()
    d = _desc(a, KH=1, KW=1, pad=0, Cout=128, out_ld=128, res=a, res_ld=64)
    assert lib.cvcs_conv2d(C.byref(d), None) == -1 and b"res view" in lib.cvcs_last_error()
    d = _desc(a, KH=1, KW=1, pad=0, Cout=64, res=a, res_ld=64)
    assert lib.cvcs_conv2d(C.byref(d), None) == -1 and b"residual epilogue needs" in lib.cvcs_last_error()
    d = _desc(a, KH=1, KW=1, pad=0, in2=a, in2_ld=32, Cin2=24)
    assert lib.cvcs_conv2d(C.byref(d), None) == -1 and b"in2 view" in lib.cvcs_last_error()
    d = _desc(a, KH=1, KW=1, pad=0, Cout=64, mask=a, mask_ld=64)
    assert lib.cvcs_conv2d(C.byref(d), None) == -1 and b"mask" in lib.cvcs_last_error()
    # Gram matrix: channels in multiples of 64
    assert lib.cvcs_gram_workspace_floats(1024, 48) == -1 and lib.cvcs_gram_workspace_floats(0, 64) == -1
    assert lib.cvcs_gram_workspace_floats(1024, 64) > 0
    assert lib.cvcs_gram(a, 48, 1024, 48, a, a, a, None) == -1 and b"multiple of 64" in lib.cvcs_last_error()
    assert lib.cvcs_bn_gram_finalize(a, a, a, 256, 48, 100, a, a, a, a, 0.1, 1e-5, a, a, a, a, a, None) == -1
    assert lib.cvcs_bn_gram_bwd(a, a, a, a, a, 250, 64, 100, a, a, a, a, a, a, a, a, None) == -1 and b"C=250" in lib.cvcs_last_error()
    assert lib.cvcs_bn_gram_mmat_workspace_floats(256, 40) == -1
    # fused head + cross-entropy: bf16, 16 / 32 / 64 channels, at most 32 classes
    args = lambda C_=16, NC=16, dt=_lib.BF16: (a, C_, 1, 8, 8, C_, a, None, a, NC, a, 1, None, 0, 1.0, a, C_, a, a, a, dt, None)   # noqa: E731
    assert lib.cvcs_head_ce(*args(dt=_lib.F32)) == -1 and b"bf16" in lib.cvcs_last_error()
    assert lib.cvcs_head_ce(*args(C_=48)) == -1 and b"C=48" in lib.cvcs_last_error()
    assert lib.cvcs_head_ce(*args(NC=40)) == -1 and b"NC=40" in lib.cvcs_last_error()
    assert lib.cvcs_head_ce_rows(1 << 20) == 1024


def test_abi14_entry_points_reject_bad_arguments(lib):
    """second half of round 4: ReLU bit masks, the folded two-source tail operand, the pooled gradient source of the tail backward, the row query of the
    thin kernel's fused BatchNorm-backward reduce"""
    keep, a = _buf(1 << 16)
    # bit masks belong to bf16 1x1 launches of the taps kernel; the writer goes with a ReLU, the reader excludes the tensor mask
    d = _desc(a, mask_bits=a)
    assert lib.cvcs_conv2d(C.byref(d), None) == -1 and b"mask_bits" in lib.cvcs_last_error()
    d = _desc(a, KH=1, KW=1, pad=0, Cout=128, out_ld=128, mask_bits=a, mask=a, mask_ld=128)
    assert lib.cvcs_conv2d(C.byref(d), None) == -1 and b"mask" in lib.cvcs_last_error()
    d = _desc(a, KH=1, KW=1, pad=0, Cout=128, out_ld=128, mask_bits_out=a)          # no ReLU
    assert lib.cvcs_conv2d(C.byref(d), None) == -1 and b"mask_bits_out goes with the ReLU" in lib.cvcs_last_error()
    d = _desc(a, KH=1, KW=1, pad=0, Cout=64, relu=1, mask_bits_out=a)               # not a taps shape
    assert lib.cvcs_conv2d(C.byref(d), None) == -1 and b"mask_bits_out" in lib.cvcs_last_error()
    # [scale3 W3 | scaled Wd]: whole 16-byte chunks per row, aligned operands
    assert lib.cvcs_bn_gram_fold(a, a, a, 60, a, a, a, 64, 256, a, a, None) == -1 and b"m1=60" in lib.cvcs_last_error()
    assert lib.cvcs_bn_gram_fold(a, a, a, 64, None, a, a, 64, 256, a, a, None) == -1 and b"null" in lib.cvcs_last_error()
    assert lib.cvcs_bn_gram_fold(a + 8, a, a, 64, a, a, a, 64, 256, a, a, None) == -1 and b"alignment" in lib.cvcs_last_error()
    # the tail backward with a pooled gradient source: the arg-max indices need the pooled gradient
    t = _lib.TailBwdDesc()
    t.dz, t.dz_ld, t.B, t.H, t.W, t.C, t.dtype = a, 64, 1, 8, 8, 64, 1
    t.y[0], t.y_ld[0], t.mean[0], t.invstd[0], t.part_dz, t.part_dzx[0] = a, 64, a, a, a, a
    assert lib.cvcs_relu_bwd_sum_bn(C.byref(t), None) == -1 and b"a gradient" in lib.cvcs_last_error()       # neither g[0] nor a pooled source
    t.pool_idx = a
    assert lib.cvcs_relu_bwd_sum_bn(C.byref(t), None) == -1 and b"pool_idx needs pool_g" in lib.cvcs_last_error()
    # row query of a thin data-gradient launch that carries the BatchNorm-backward reduce: one row per resident workgroup, never more than the tiles
    d = _desc(a, Cin=32, in_ld=32, Cout=16, out_ld=16, H=16, W=64, Ho=16, Wo=64, bwd_y=16, bwd_mode=0)
    rows = lib.cvcs_conv_stat_rows(C.byref(d))
    assert 1 <= rows <= 4              # (4 tiles of 4 x 64 pixels; without a device the resident-workgroup query falls back to 1024)
    d.bwd_y = None
    assert lib.cvcs_conv_stat_rows(C.byref(d)) == 16      # forward statistics: one row per tile row
    # a conv / weight gradient over the bilinear x2 up-sampling of its stored input: the thin 32 -> 16 instance only
    d = _desc(a, in_up2=1)                                              # 32 -> 64: not a thin shape
    assert lib.cvcs_conv2d(C.byref(d), None) == -1 and b"in_up2" in lib.cvcs_last_error()
    d = _desc(a, Cout=16, out_ld=16, H=15, W=64, Ho=15, Wo=64, in_up2=1)   # odd height
    assert lib.cvcs_conv2d(C.byref(d), None) == -1 and b"in_up2" in lib.cvcs_last_error()
    w = _lib.WgradDesc()
    w.x, w.x_ld, w.B, w.H, w.W, w.Cin, w.dy, w.dy_ld, w.Ho, w.Wo, w.Cout = a, 64, 1, 16, 64, 64, a, 64, 16, 64, 64
    w.KH, w.KW, w.stride, w.pad, w.dw, w.Cin_real, w.workspace, w.dtype, w.x_up2 = 3, 3, 1, 1, a, 64, a, 1, 1
    assert lib.cvcs_conv2d_wgrad(C.byref(w), None) == -1 and b"x_up2" in lib.cvcs_last_error()

