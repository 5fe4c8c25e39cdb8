"""Host logic of the launch plans, without a GPU: every network's forward (train + eval) and backward plan is recorded with CPU
tensors standing in for the device buffers.  Each C-ABI call then runs its full argument validation (shapes, leading dimensions,
alignment, workspace sizes, channel-count rules) and fails only at the launch itself ("no device"), which this test tolerates;
any CVCS_EINVAL, Python-side shape assertion or mis-sized buffer fails the test.  Nothing is computed here - numerical parity is
the job of the `-m gpu` tests."""
from collections import OrderedDict

import pytest
import torch

from cvcs_amd import _lib, nets, ops


@pytest.fixture()
def no_device(monkeypatch):
    if torch.cuda.is_available():
        pytest.skip("a device is present: the launches would run")
    failed = [0]

    def check(rc, what=""):
        if rc == 0:
            return
        msg = _lib._load().cvcs_last_error().decode(errors="replace")
        if rc == -2 and "launch failed" in msg:      # CVCS_EHIP from CVCS_CHECK_LAUNCH: the arguments were accepted
            failed[0] += 1
            return
        raise _lib.CvcsError(f"{what} failed ({rc}): {msg}")
    monkeypatch.setattr(_lib, "C_REPLAY", False)        # (the C replay stops at the first refused launch; here every launch is refused: no device)
    monkeypatch.setattr(ops, "_stream", lambda: 0)
    monkeypatch.setattr(ops, "check", check)
    monkeypatch.setattr(_lib, "check", check)
    return failed


def _bind_on_cpu(net, dtype):
    params = net._flat_order(OrderedDict(net.named_parameters()))
    total = sum(nets._align4(p.numel()) for p in params.values())
    flat, fg, off = torch.zeros(total), torch.zeros(total), 0
    P, G = OrderedDict(), OrderedDict()
    for k, p in params.items():
        n = p.numel()
        P[k], G[k] = flat[off:off + n].view(p.shape), fg[off:off + n].view(p.shape)
        off += nets._align4(n)
    eng = net._build_engine("cpu")
    eng.bind(P, G, {k: v for k, v in net.named_buffers() if v.dtype == torch.float32})
    return eng


CASES = [("SwinTUperNet", {}, 4, 64), ("SwinTUperNet", {}, 2, 224), ("SwinBUperNet", {}, 2, 96), ("Resnet18Unet", {}, 2, 64), ("Resnet50Unet", {}, 2, 96),
         ("Resnet50Unet", {"decoder_norm": "gn_silu"}, 2, 64), ("DeepLabV3Plus", {}, 2, 64), ("DeepLabv3Resnet101", {}, 2, 64),
         ("SegformerMod", {}, 2, 64), ("SegformerMod", {"stochastic": False}, 2, 256), ("SegformerMod", {"variant": "b3"}, 2, 96),
         ("DeepLabV3MobileNet", {}, 2, 64), ("DeepLabV3MobileNet", {}, 3, 224)]


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("cls,kw,B,S", CASES)
def test_every_launch_of_the_plan_passes_the_c_abi_argument_checks(no_device, cls, kw, B, S, precision):
    net = getattr(nets, cls)(5, precision, **kw)
    eng = _bind_on_cpu(net, nets.PRECISIONS[precision])
    x = torch.zeros(B, 3, S, S)
    out = eng.forward(x, True)
    assert tuple(out.shape) == (B, 5, S, S)
    eng.backward(torch.zeros_like(out))
    eng.forward(x, False)
    fwd, bwd = len(eng._rec["fwd_train"].items), len(eng._rec["bwd"].items)
    assert fwd > 20 and bwd > fwd and no_device[0] >= fwd + bwd


@pytest.mark.parametrize("cls", ["SwinTUperNet", "DeepLabV3Plus"])
def test_one_tile_batches_are_refused_in_train_mode_like_torch(no_device, cls):
    """a BatchNorm over a 1x1 map of ONE image has no variance: torch raises ValueError, so do these engines (eval mode is fine)"""
    net = getattr(nets, cls)(5, "fp32")
    eng = _bind_on_cpu(net, torch.float32)
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        eng.forward(torch.zeros(1, 3, 64, 64), True)
    assert tuple(eng.forward(torch.zeros(1, 3, 64, 64), False).shape) == (1, 5, 64, 64)


def _idx3(x, d1, d2, s0, s1, s2):
    return (x // (d1 * d2)) * s0 + ((x // d2) % d1) * s1 + (x % d2) * s2


@pytest.mark.parametrize("cls,kw", [("DeepLabV3MobileNet", {}), ("SegformerMod", {})])
def test_gather_tables_cover_every_master_element_exactly_once(cls, kw):
    """host logic of the table-driven weight gather / gradient scatter (cvcs_gather_item, include/cvcs_hip.h): for every scatter item the
    map (row, column) -> master index must be a bijection onto the master tensor (every gradient element written once, none twice), and every
    gather item must stay inside its source and its padded destination.  Evaluated from the item dictionaries the engines keep."""
    import numpy as np
    net = getattr(nets, cls)(5, "bf16", **kw)
    eng = _bind_on_cpu(net, torch.bfloat16)
    tables = eng._keep
    gath, scat = tables[0], tables[1]
    assert len(gath) > 20 and len(scat) > 10
    seen = {}
    for it in scat:
        rv2, cv2 = it.get("rv2", 0), it.get("cv2", 0)
        r = np.arange(it["Rv"]); c = np.arange(it["Cv"])
        r = r[(r % it["row"][1] < rv2)] if rv2 else r
        c = c[(c % it["col"][1] < cv2)] if cv2 else c
        idx = (it.get("base", 0) + _idx3(r, *it["row"])[:, None] + _idx3(c, *it["col"])[None, :]).ravel()
        n = it["src"].numel()
        assert idx.min() >= 0 and idx.max() < n, "scatter item writes outside its master tensor"
        assert len(np.unique(idx)) == len(idx), "scatter item writes a master element twice"
        key = it["src"].data_ptr()
        seen.setdefault(key, [n, 0])[1] += len(idx)
        assert it["dst"].numel() >= it["R"] * it["Cp"] and it["Rv"] <= it["R"] and it["Cv"] <= it["Cp"]
    if cls == "DeepLabV3MobileNet":      # every backbone gradient goes through exactly one item: the items of a tensor cover it completely
        assert all(n == hit for n, hit in seen.values()), [v for v in seen.values() if v[0] != v[1]][:3]
    for it in gath:
        rv2, cv2 = it.get("rv2", 0), it.get("cv2", 0)
        r = np.arange(it["Rv"]); c = np.arange(it["Cv"])
        r = r[(r % it["row"][1] < rv2)] if rv2 else r
        c = c[(c % it["col"][1] < cv2)] if cv2 else c
        if len(r) == 0 or len(c) == 0:
            continue
        idx = it.get("base", 0) + _idx3(r, *it["row"])[:, None] + _idx3(c, *it["col"])[None, :]
        assert idx.min() >= 0 and idx.max() < it["src"].numel(), "gather item reads outside its master tensor"
        assert it["dst"].numel() >= it["R"] * it["Cp"]
