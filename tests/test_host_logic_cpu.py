"""CPU-only: the host-side mirror of the reference surface that needs no GPU - metrics (S/utils.py:311-403), scheduler
(S/utils.py:208-221), class weights (S/dataset.py:360-384), palette converter (S/converters.py), the GID-15 chunk loader
on a tiny on-disk dataset (S/dataset.py:228-387), parameter layout (S/nets.py) - against the oracle, torch and sklearn."""
import os

import numpy as np
import pytest
import torch

from cvcs_amd import converters, dataset, nets, optim, utils
from oracle import unet_oracle as O


def test_metrics_match_oracle_and_sklearn():
    from sklearn.metrics import jaccard_score, precision_score, recall_score
    rng = np.random.default_rng(0)
    K = 16
    t = rng.integers(0, 12, 20000)          # classes 12..15 never occur in the target -> excluded from the means
    p = np.where(rng.random(20000) < 0.7, t, rng.integers(0, K, 20000))
    conf = torch.from_numpy(O.confusion_matrix(p, t, K, ignore_index=0))
    m = utils.print_metrics(conf, silent=True)
    o = O.metrics(conf.numpy())
    for k in ("mIoU", "precision_score", "recall_score", "dice_score", "oa_score"):
        assert m[k] == pytest.approx(o[k], rel=1e-6), k
    keep = t != 0
    present = sorted(set(t[keep]))
    sk = jaccard_score(t[keep], p[keep], labels=present, average="macro")
    assert m["mIoU"] == pytest.approx(sk, rel=1e-6)                      # mean over classes present in the target
    assert m["recall_score"] == pytest.approx(recall_score(t[keep], p[keep], labels=present, average="macro"), rel=1e-6)
    vals, excluded = utils.IoU(conf, mean=False, return_excluded=True)
    assert excluded == [0, 12, 13, 14, 15] and len(vals) == K
    pred_present = sorted(set(p[keep]))
    assert m["precision_score"] == pytest.approx(
        precision_score(t[keep], p[keep], labels=pred_present, average="macro", zero_division=0), rel=1e-6)


def test_polynomial_lr_is_torchs_schedule_and_round_trips():
    class Opt:
        lr = 0.005
    for total, power in ((20, 1.0), (7, 2.0)):
        p = torch.nn.Parameter(torch.zeros(1))
        topt = torch.optim.SGD([p], lr=0.005)
        ts = torch.optim.lr_scheduler.PolynomialLR(topt, total_iters=total, power=power)
        o = Opt(); o.lr = 0.005
        s = optim.PolynomialLR(o, total_iters=total, power=power)
        for _ in range(total + 4):
            assert o.lr == pytest.approx(topt.param_groups[0]["lr"], rel=1e-12, abs=1e-18)
            topt.step(); ts.step(); s.step()
        o2 = Opt(); o2.lr = 0.005
        s2 = optim.PolynomialLR(o2, total_iters=total, power=power)
        s2.load_state_dict(s.state_dict())
        assert o2.lr == o.lr and s2.last_epoch == s.last_epoch


def test_class_weights_match_the_oracle():
    counts = torch.tensor([50.0, 10.0, 0.0, 70.0, 20.0])
    for ib in (False, True):
        np.testing.assert_allclose(dataset.class_weights_from_counts(counts, ib).numpy(),
                                   O.class_weights(counts.tolist(), ignore_background=ib), rtol=1e-6)


def test_converter_round_trip_and_palette(golden_dir):
    g = np.load(os.path.join(golden_dir, "converter_ref.npz"))     # written by the reference's own GID15Converter
    conv = converters.GID15Converter()
    pal = conv.palette_u8().numpy()
    assert pal.shape == (16, 3) and (pal[g["labels"]] == g["colors"]).all()
    assert np.allclose(conv.iconvert(torch.arange(16).reshape(4, 4)).numpy(), g["iconvert_4x4"])
    assert all(conv.color_to_label[tuple(int(v) for v in c)] == int(l) for c, l in zip(g["colors"], g["labels"]))
    assert torch.equal(conv.iconvert(torch.tensor([[99]]))[0, 0], torch.ones(3))     # unknown label stays white


def test_gid15_loader_on_a_tiny_dataset(tmp_path):
    """the reference's directory layout (Image__8bit_NirRGB / Annotation__index), patch_size 224: tiles-per-image,
    chunking, specify(), whole-image coverage of the crops and the class-weight scan"""
    from PIL import Image
    rng = np.random.default_rng(1)
    os.makedirs(tmp_path / "Image__8bit_NirRGB"); os.makedirs(tmp_path / "Annotation__index")
    imgs, masks = [], []
    for i in range(3):
        im = rng.integers(0, 256, (460, 700, 3), dtype=np.uint8)
        mk = rng.integers(0, 5, (460, 700), dtype=np.uint8)
        Image.fromarray(im).save(tmp_path / "Image__8bit_NirRGB" / f"t{i}.png")
        Image.fromarray(mk).save(tmp_path / "Annotation__index" / f"t{i}.png")
        imgs.append(im); masks.append(mk)
    ld = dataset.make_loader(str(tmp_path), 2, 224, 5)
    assert ld.tpi == 2 * 3 and len(ld) == 2 and ld.chunks == [[0, 1], [2]]
    chunk = ld.get_iterable_chunk(1)
    tiles = list(chunk)
    assert len(tiles) == 6
    # every tile is an exact crop of image 2 at a tile-aligned origin, and together they cover all six origins
    seen = set()
    for img, lab, _, _ in tiles:
        assert img.shape == (3, 224, 224) and img.dtype == torch.uint8 and lab.shape == (224, 224)
        hits = [(ty, tx) for ty in range(2) for tx in range(3)
                if np.array_equal(img.permute(1, 2, 0).numpy(), imgs[2][ty * 224:(ty + 1) * 224, tx * 224:(tx + 1) * 224])]
        assert len(hits) == 1
        ty, tx = hits[0]
        assert np.array_equal(lab.numpy(), masks[2][ty * 224:(ty + 1) * 224, tx * 224:(tx + 1) * 224])
        seen.add(hits[0])
    assert len(seen) == 6
    w = ld.get_class_weights(5, ignore_background=True)
    counts = np.bincount(np.concatenate([m.reshape(-1) for m in masks]), minlength=5)
    np.testing.assert_allclose(w.numpy(), O.class_weights(counts.tolist(), ignore_background=True), rtol=1e-5)
    ld.specify([2])
    assert ld.chunks == [[2]]
    with pytest.raises(AssertionError):
        dataset.make_loader(str(tmp_path), 2, 200, 5)          # S/dataset.py:268: only 224 / 256 / 512


def test_synthetic_loader_is_deterministic_and_exercises_ignore_index():
    a = dataset.make_loader("synthetic:2:3", 1, 64, 6, seed=5)
    b = dataset.make_loader("synthetic:2:3", 1, 64, 6, seed=5)
    ta, tb = list(a.get_iterable_chunk(1)), list(b.get_iterable_chunk(1))
    assert len(ta) == 3 and all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(ta, tb))
    labs = torch.cat([t[1].reshape(-1) for c in range(len(a)) for t in a.get_iterable_chunk(c)])
    assert int(labs.max()) < 6 and (labs == 0).any() and (labs > 0).any()


@pytest.mark.parametrize("variant,NC,count", [("Unetv2", 16, 31044496), ("Unet", 16, 34526096)])
def test_parameter_count_and_state_dict_layout(variant, NC, count):
    """31 044 496 / 34 526 096 parameters at NC = 16; state_dict keys and shapes are the reference's
    (the oracle's parameter table is pinned by the reference's own key list in tests/golden)."""
    net = (nets.Urnetv2 if variant == "Unetv2" else nets.Urnet)(NC)
    assert utils.count_params(net) == count
    sd = net.state_dict()
    ref = O.init_params(variant, NC, seed=0)
    assert set(k for k in sd if not k.endswith("num_batches_tracked")) == set(ref)
    assert all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in ref)


def test_host_only_c_abi_sizing_functions():
    """the sizing entry points of the C-ABI are pure host code (no device call): partial-statistics rows of a conv launch
    (include/cvcs_hip.h cvcs_conv_stat_rows), split-K slices of a weight gradient (cvcs_wgrad_slices), the row counts of
    the two-pass reductions - checked for every layer shape of the bench workload and for ragged ones."""
    import ctypes as C
    from cvcs_amd import _lib
    lib = _lib.lib()
    for dtype, per_tile in ((_lib.BF16, 1), (_lib.F32, 4)):
        for (B, H, W, Cin, Cout) in ((32, 512, 512, 64, 64), (32, 256, 256, 128, 128), (2, 40, 24, 64, 128), (1, 17, 33, 64, 64)):
            d = _lib.ConvDesc()
            d.B, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = B, H, W, Cin, H, W, Cout
            d.KH, d.KW, d.stride, d.pad, d.dil, d.dtype = 3, 3, 1, 1, 1, dtype
            tiles = B * -(-H // 16) * -(-W // 16)
            assert lib.cvcs_conv_stat_rows(C.byref(d)) == tiles * per_tile   # one row per 16x16 tile (bf16) / per wave row group (f32)
        d = _lib.ConvDesc()
        d.B, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = 2, 8, 8, 128, 8, 8, 256
        d.KH, d.KW, d.stride, d.pad, d.dil, d.dtype = 1, 1, 1, 0, 1, dtype
        if dtype == _lib.BF16:   # 1x1 with 128 | Cout: the non-overlapping-taps kernel, four 64-pixel wave blocks per 256-pixel tile
            assert lib.cvcs_conv_stat_rows(C.byref(d)) == 4 * -(-2 * 8 * 8 // 256)
        else:                    # generic kernel: two rows per 128-pixel tile
            assert lib.cvcs_conv_stat_rows(C.byref(d)) == 2 * -(-2 * 8 * 8 // 128)
    bad = _lib.ConvDesc()
    assert lib.cvcs_conv_stat_rows(C.byref(bad)) < 0
    # every layer of Unetv2 at B=32, S=512 (3x3 convs and the four 2x2/s2 ConvTranspose gradients): at least one slice, never
    # more slices than K-tiles, and the workspace of the deepest layer stays small (few slices where there are many tiles)
    for (S, Cin, Cout, K, stride) in ((512, 32, 64, 3, 1), (512, 64, 64, 3, 1), (512, 128, 64, 3, 1), (256, 128, 128, 3, 1),
                                      (64, 512, 512, 3, 1), (32, 1024, 1024, 3, 1), (32, 512, 1024, 2, 2), (256, 64, 128, 2, 2)):
        n = lib.cvcs_wgrad_slices(32, S, S, Cout, Cin, K, K, stride)
        assert 1 <= n <= 32 * S * S // 32
        if Cin * Cout >= 512 * 512:
            assert n <= 8
    assert lib.cvcs_wgrad_slices(0, 8, 8, 64, 64, 3, 3, 1) < 0 and lib.cvcs_wgrad_slices(1, 8, 8, 8, 64, 3, 3, 1) < 0
    for M in (1, 255, 256, 257, 1 << 23):
        assert lib.cvcs_bn_bwd_rows(M) == min(1024, max(1, -(-M // 64)))
        assert lib.cvcs_head_bwd_rows(M) == min(1024, max(1, -(-M // 64)))
        assert lib.cvcs_ce_workspace_floats(M) == 2 + 2 * min(4096, max(1, -(-M // 1024)))


def test_c_replay_trampoline_passes_register_stack_and_float_arguments_in_place():
    """cvcs_replay calls every launch entry point through ONE 28-integer + 8-float prototype (x86-64 System V): arguments 1-6 travel in general
    registers, the rest in stack slots, floats in xmm registers.  Host-side argument checks that echo a value prove each class lands where the
    callee reads it (no GPU needed: the calls are refused before any launch)."""
    import ctypes as C
    from cvcs_amd import _lib
    h = _lib._load()
    assert h.cvcs_sizeof_call() == C.sizeof(_lib.Call)

    def one(fn, ints, flts=()):
        arr = (_lib.Call * 1)()
        arr[0].fn = C.cast(fn, C.c_void_p).value
        arr[0].nint, arr[0].nflt = len(ints), len(flts)
        for k, v in enumerate(ints):
            arr[0].i[k] = v
        for k, v in enumerate(flts):
            arr[0].f[k] = v
        failed = C.c_int(-1)
        rc = h.cvcs_replay(arr, 1, None, C.byref(failed))
        return rc, failed.value, h.cvcs_last_error().decode()

    # 4th integer argument (a register): cvcs_layernorm_fwd(x, x_ld, M, C, gamma, beta, eps, out, out_ld, mean, invstd, dtype)
    rc, idx, msg = one(h.cvcs_layernorm_fwd, [4096, 16, 7, 13, 4096, 4096, 4096, 16, 4096, 4096, 1], [1e-5])
    assert rc == -1 and idx == 0 and "C=13" in msg, msg
    # 9th integer argument (a stack slot): cvcs_sr_attention_fwd(q, q_ld, kv, kv_ld, B, N, Nk, heads, head_dim, ...)
    rc, idx, msg = one(h.cvcs_sr_attention_fwd, [4096, 64, 4096, 128, 2, 16, 4, 2, 48, 4096, 64, 4096, 1])
    assert rc == -1 and "head_dim=48" in msg, msg
    # a float argument: cvcs_dropout(x, x_ld, M, C, out, out_ld, state, p, dtype) refuses p >= 1 but accepts p = 0.5 up to the next check
    rc, _, msg = one(h.cvcs_dropout, [4096, 8, 4, 8, 4096, 8, 4096, 1], [1.5])
    assert rc == -1 and "bad argument" in msg
    import torch
    if not torch.cuda.is_available():     # (with a device the accepted call would really launch on these dummy pointers)
        rc, _, msg = one(h.cvcs_dropout, [4096, 8, 4, 8, 4096, 8, 4096, 1], [0.5])
        assert rc != 0 and "bad argument" not in msg, msg        # past the p check: refused at the launch itself (no device here)
    # a bad record is refused by cvcs_replay itself
    arr = (_lib.Call * 1)()
    assert h.cvcs_replay(arr, 1, None, None) == -1 and "bad record" in h.cvcs_last_error().decode()


def test_dispatch_table_joins_trace_and_counter_passes_of_the_last_step(tmp_path):
    """scripts/dispatch_table.py: the launches of the LAST training step (between the last two sgd_kernel dispatches) of a kernel trace, joined by
    dispatch order with separate FETCH_SIZE / WRITE_SIZE passes of the same command (gfx950 correction: FETCH_SIZE doubled, both in KiB)"""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = ["pack_k", "conv_k<unsigned short, 64>", "sgd_kernel"]
    trace = ['"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id","Start_Timestamp","End_Timestamp",'
             '"LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Workgroup_Size_X","Workgroup_Size_Y","Workgroup_Size_Z","Grid_Size_X","Grid_Size_Y","Grid_Size_Z"']
    did = 0
    for step in range(3):
        for j, n in enumerate(names):
            did += 1
            t0 = did * 100000
            trace.append(f'"KERNEL_DISPATCH","Agent 2",1,0,1,{did},{j},"{n}",{did},{t0},{t0 + 1000 * (j + 1) * (step + 1)},0,0,8,0,32,256,1,1,{256 * 4},2,1')

    def counters(name, base):
        rows = ['"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size",'
                '"Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"']
        d = 0
        for step in range(2):
            for j, n in enumerate(names):
                d += 1
                rows.append(f'{d},{d},"Agent 2",1,1,1,1024,{j},"{n}",256,0,0,8,0,32,"{name}",{base * (j + 1)},0,1')
        return rows
    (tmp_path / "t.csv").write_text("\n".join(trace) + "\n")
    (tmp_path / "f.csv").write_text("\n".join(counters("FETCH_SIZE", 1000.0)) + "\n")
    (tmp_path / "w.csv").write_text("\n".join(counters("WRITE_SIZE", 500.0)) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dispatch_table.py"), str(tmp_path / "t.csv"), str(tmp_path / "f.csv"), str(tmp_path / "w.csv")],
                         capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0].startswith("# 3 dispatches in the last step; pmc passes aligned: True")
    assert "pack_k" in out[1] and "4x2x1" in out[1] and "3.0 us" in out[1]                 # step 2 (the last): 1000 * 1 * 3 ns
    assert "conv_k<bf16, 64>" in out[2] and f"rd {2 * 2000 * 1024 / 1e6:8.1f} MB" in out[2] and f"wr {1000 * 1024 / 1e6:8.1f} MB" in out[2]
    assert out[-1].startswith("# total 0.02 ms")


def test_wide_input_weight_gradient_swizzle_spreads_a_transposed_read_over_all_banks():
    """wgrad_thin_wide_kernel (csrc/wgrad.hip): a halo pixel is 256 bytes = the LDS bank period, so chunk c of halo pixel p is stored at chunk
    c ^ key(p), key(p) = 2 * ((p & 3) | (((p >> 3) & 1) << 2)).  Restated here: (a) the map is a bijection of a pixel's sixteen 16-byte chunks
    (the DMA source permutation and the read address are inverse to each other); (b) for every tap offset and wave, the 64 lanes of one
    ds_read_b64_tr_b16 (pixels base + 8 fg + q, q = 0..3, fg = 0..3; two adjacent chunks per pixel, 8 bytes each) land on each of the sixteen
    16-byte bank groups exactly four times (512 bytes over 256 bytes of banks: two-way, the minimum) - un-swizzled they all fall on two groups"""
    key = lambda p: 2 * ((p & 3) | (((p >> 3) & 1) << 2))   # noqa: E731
    for p in range(64):
        assert sorted(c ^ key(p) for c in range(16)) == list(range(16))
    for base in range(0, 6 * 34):                  # any halo pixel a tap / tile row can start from
        for c0 in range(0, 16, 2):                 # (wave, j): the even chunk of a 16-channel block
            for second in (0, 4):                  # the two transposed reads of a fragment
                hits, plain = [0] * 16, [0] * 16
                for fg in range(4):
                    for q in range(4):
                        px = base + 8 * fg + q + second
                        for half in (0, 1):        # lanes pp = 0..3: chunks c0, c0 + 1, two 8-byte halves each
                            for _ in (0, 1):
                                hits[(c0 + half) ^ key(px)] += 1
                                plain[c0 + half] += 1
                assert max(hits) == 4 and min(hits) == 4, (base, c0, hits)
                assert max(plain) == 32

