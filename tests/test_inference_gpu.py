"""Whole-image inference (border correction + stitching) and checkpoint compatibility on the GPU."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cvcs_amd import nets, ops, utils  # noqa: E402
from cvcs_amd.converters import GID15Converter  # noqa: E402
from cvcs_amd.inference import segment_image  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402

DEV = "cuda:0"


def test_crop_tiles_matches_slicing():
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (3, 70, 100), dtype=torch.uint8, generator=g)
    p, S = 32, 48
    Wn = 100 // p
    m = (S - p) // 2
    dst = torch.empty(4, 3, S, S, dtype=torch.uint8, device=DEV)
    ops.crop_tiles(img.to(DEV), dst, 1, Wn, p)          # tiles 1..4 of a 2 x 3 grid
    torch.cuda.synchronize()
    padded = F.pad(img, (m, S, m, S))                    # zero ring
    for t in range(4):
        ty, tx = divmod(1 + t, Wn)
        ref = padded[:, ty * p:ty * p + S, tx * p:tx * p + S]
        assert torch.equal(dst[t].cpu(), ref), t


@pytest.mark.parametrize("bc", [None, 48])
def test_segment_image_matches_oracle(bc):
    """utils.inference + stitching (S/utils.py:145-171, S/inference.py:40-57) with centred border correction."""
    NC, p = 5, 32
    S = bc or p
    m = (S - p) // 2
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 256, (3, 64 + 7, 96 + 5), dtype=torch.uint8, generator=g)   # 2 x 3 tiles + ragged edge
    net = nets.Urnetv2(NC, "fp32")
    net.load_state_dict(O.init_params("Unetv2", NC, seed=3), strict=False)
    net = net.to(DEV)
    rgb, labels = segment_image(net, img.to(DEV), p, bc, batch=4)
    torch.cuda.synchronize()
    assert labels.shape == (64, 96) and rgb.shape == (64, 96, 3)
    params = O.init_params("Unetv2", NC, seed=3)
    padded = F.pad(img, (m, S, m, S)).float()
    pal = GID15Converter().palette_u8()
    agree, total = 0, 0
    for ty in range(2):
        for tx in range(3):
            win = padded[None, :, ty * p:ty * p + S, tx * p:tx * p + S]
            with torch.no_grad():
                ref = O.unet_forward(params, win, "Unetv2", train=False)[0, :, m:m + p, m:m + p]
            top2 = torch.topk(ref, 2, dim=0).values
            decided = (top2[0] - top2[1]) > 1e-3 * ref.abs().max()
            got = labels[ty * p:(ty + 1) * p, tx * p:(tx + 1) * p].cpu().long()
            assert torch.equal(got[decided], ref.argmax(0)[decided])
            agree += (got == ref.argmax(0)).sum().item(); total += p * p
            assert torch.equal(rgb[ty * p:(ty + 1) * p, tx * p:(tx + 1) * p].cpu(), pal[got])
    assert agree / total > 0.999


def test_checkpoint_round_trip_and_reference_layout(tmp_path):
    """save_model / load_checkpoint (S/utils.py:128-142, 282-299): reference keys; a checkpoint holding a reference-style
    state dict under a DataParallel `module.` prefix loads; weights survive the round trip bit for bit."""
    NC = 5
    net = nets.Urnetv2(NC, "fp32").to(DEV)
    opt, sched = utils.load_optimizer({"opt": "SGD2", "epochs": 2}, net)
    img, lab = O.synthetic_tiles(2, 32, NC, seed=1)
    crit = utils.CrossEntropyLoss(ignore_index=0)
    net.train()
    loss = crit(net(img.to(DEV), None), lab.to(DEV))
    opt.zero_grad(); loss.backward(); opt.step()
    utils.save_model(0, net, opt, sched, [loss.item()], [], [], [], [], [], 2, str(tmp_path), "SGD2")
    ck = torch.load(os.path.join(tmp_path, "checkpoint1"), map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "training_loss_values",
                       "validation_loss_values", "batch_size", "macro_precision", "weighted_precision", "conf_flat",
                       "conf_normalized", "optimizer"}
    assert int(ck["model_state_dict"]["encode1.0.layer.1.num_batches_tracked"]) == 1
    net2 = nets.Urnetv2(NC, "fp32").to(DEV)
    utils.load_checkpoint({"load_checkpoint": os.path.join(tmp_path, "checkpoint1")}, net2)
    for (k, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k
    # reference-style checkpoint: torch-CPU tensors with a `module.` prefix
    ref_sd = {"module." + k: v for k, v in O.init_params("Unetv2", NC, seed=9).items()}
    torch.save({"model_state_dict": ref_sd}, os.path.join(tmp_path, "ref_ckpt"))
    net3 = nets.Urnetv2(NC, "fp32")
    missing = utils.load_checkpoint  # noqa: F841
    sd = {k[len("module."):]: v for k, v in ref_sd.items()}
    net3.load_state_dict(sd, strict=False)
    net3 = net3.to(DEV)
    net3.eval()
    with torch.no_grad():
        out = net3(img.to(DEV), None).cpu()
        ref = O.unet_forward(O.init_params("Unetv2", NC, seed=9), img.float(), "Unetv2", train=False)
    assert (out - ref).abs().max() < 1e-3 * ref.abs().max()


def test_device_chunk_tile_producer_matches_reference_crops():
    """tile producer (S/dataset.py:136-172): resident images + gather launches == the eager Python crops, incl. a
    randomly shifted grid (zero fill past the border) and the class-weight histogram (S/dataset.py:346-384)."""
    from cvcs_amd.dataset import DeviceChunk, class_weights_from_counts, device_class_weights
    g = torch.Generator().manual_seed(5)
    images = [torch.randint(0, 256, (3, 96, 130), dtype=torch.uint8, generator=g) for _ in range(2)]
    masks = [torch.randint(0, 6, (96, 130), dtype=torch.uint8, generator=g) for _ in range(2)]
    dimg, dmsk = [i.to(DEV) for i in images], [m.to(DEV) for m in masks]
    for shift in (False, True):
        ch = DeviceChunk(dimg, dmsk, 32, batch_size=5, random_shift=shift, shuffle=True, seed=1)
        assert len(ch.items) == 2 * 3 * 4 and len(ch) == 5
        seen = 0
        for b, (im, mk, _, _) in enumerate(ch):
            assert im.dtype == torch.uint8 and im.is_cuda and im.shape[1:] == (3, 32, 32) and mk.shape[1:] == (32, 32)
            for j in range(im.shape[0]):
                i, y, x = ch.items[b * 5 + j]
                pad = 32
                ref_i = F.pad(images[i], (pad, pad, pad, pad))[:, y + pad:y + pad + 32, x + pad:x + pad + 32]
                ref_m = F.pad(masks[i], (pad, pad, pad, pad))[y + pad:y + pad + 32, x + pad:x + pad + 32]
                assert torch.equal(im[j].cpu(), ref_i) and torch.equal(mk[j].cpu(), ref_m)
                seen += 1
        assert seen == 24
    counts = sum(torch.bincount(m.reshape(-1).long(), minlength=6) for m in masks).float()
    for ib in (False, True):
        assert torch.equal(device_class_weights(dmsk, 6, ib), class_weights_from_counts(counts, ib))


@pytest.mark.parametrize("M,NC", [(1, 5), (3, 3), (4, 2), (7, 16), (16, 4)])
def test_vote_labels_matches_torch_mode(M, NC):
    """per-pixel majority vote (Ensemble.forward, S/utils.py:504-507): torch.mode semantics, ties -> smallest label."""
    g = torch.Generator().manual_seed(M * 31 + NC)
    stack = torch.randint(0, NC, (M, 2, 37, 53), generator=g, dtype=torch.uint8)
    out = torch.empty((2, 37, 53), dtype=torch.uint8, device=DEV)
    ops.vote_labels(stack.to(DEV), out)
    want, _ = torch.mode(stack.long(), dim=0)
    assert torch.equal(out.cpu().long(), want)


def test_label_confusion_matches_bincount():
    g = torch.Generator().manual_seed(5)
    K = 16
    pred = torch.randint(0, K, (3, 41, 29), generator=g, dtype=torch.uint8)
    tgt = torch.randint(0, K, (3, 41, 29), generator=g, dtype=torch.int64)
    conf = torch.zeros(K, K, dtype=torch.int64, device=DEV)
    ops.label_confusion(pred.to(DEV).view(-1), tgt.to(DEV).view(-1), 0, K, conf)
    ops.label_confusion(pred.to(DEV).view(-1), tgt.to(torch.uint8).to(DEV).view(-1), 0, K, conf)   # accumulates
    keep = tgt != 0
    want = torch.bincount(tgt[keep] * K + pred.long()[keep], minlength=K * K).view(K, K)
    assert torch.equal(conf.cpu(), 2 * want)


def test_ensemble_votes_like_the_reference(tmp_path):
    """utils.Ensemble (S/utils.py:472-507): members loaded from checkpoints by network name, per-member argmax, per-pixel
    torch.mode; checked against the oracle's forward of the same weights, member by member and after the vote."""
    import yaml
    NC, S = 5, 64
    img, lab = O.synthetic_tiles(2, S, NC, seed=12, structured=True)
    members = {"Unetv2": 31, "Unet": 32}
    cfg = {}
    for name, seed in members.items():
        sd = dict(O.init_params(name, NC, seed=seed))
        for k in [k for k in sd if k.endswith("running_mean")]:   # a torch-written checkpoint carries the BN step counters
            sd[k.replace("running_mean", "num_batches_tracked")] = torch.tensor(0)
        torch.save({"model_state_dict": sd}, os.path.join(tmp_path, f"ck_{name}"))
        cfg[name] = os.path.join(str(tmp_path), f"ck_{name}")
    with open(os.path.join(tmp_path, "ens.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    ens = utils.load_network({"net": "Ensemble", "num_classes": NC - 1, "precision": "fp32",
                              "ensemble_config": os.path.join(str(tmp_path), "ens.yaml")}, DEV)
    assert ens.returns_logits is False and len(ens.models) == 2
    got = ens(img.to(DEV), None).cpu()
    assert got.dtype == torch.int64 and got.shape == (2, S, S)
    refs = [O.unet_forward(O.init_params(name, NC, seed=seed), img.float(), name, train=False) for name, seed in members.items()]
    want, _ = torch.mode(torch.stack([r.argmax(1) for r in refs]), dim=0)
    # undecided pixels (top-2 logits of a member within 1e-3 relative) may flip that member's vote
    decided = torch.ones_like(want, dtype=torch.bool)
    for r in refs:
        top2 = r.topk(2, dim=1).values
        decided &= (top2[:, 0] - top2[:, 1]) > 1e-3 * r.abs().max()
    assert decided.float().mean() > 0.9 and torch.equal(got[decided], want[decided])
    # single tile -> [H,W] like the reference's squeeze; eval_model takes the label path
    assert ens(img[:1].to(DEV), None).shape == (S, S)
    from cvcs_amd import dataset
    loader = dataset.make_loader("synthetic:2:2", 2, S, NC, seed=1)
    flat, _ = utils.eval_model(ens, loader, DEV, batch_size=2, ignore_background=True, num_classes=16)
    assert int(flat.compute().sum()) > 0


@pytest.mark.parametrize("precision,NC", [("fp32", 5), ("bf16", 16), ("bf16", 21)])
def test_predict_labels_equals_argmax_of_logits(precision, NC):
    """net.predict_labels (head + argmax fused, no logits tensor) == torch.argmax(net(x)) of the eval-mode logits, bit for
    bit, ties included (same order of operations as cvcs_head_fwd); label_stitch == argmax_stitch on the same tiles."""
    torch.manual_seed(7)
    net = (nets.Urnetv2 if NC != 5 else nets.Urnet)(NC, precision).to(DEV)
    net.eval()
    img, _ = O.synthetic_tiles(3, 96, NC, seed=3)
    x = img.to(DEV)
    with torch.no_grad():
        logits = net(x, None)
    lab = net.predict_labels(x)
    assert lab.dtype == torch.uint8 and lab.shape == (3, 96, 96)
    assert torch.equal(lab.long(), torch.argmax(logits, 1))
    pal = GID15Converter().palette_u8(DEV)
    p, Wn = 64, 2
    rgb_a = torch.zeros(2 * p, Wn * p, 3, dtype=torch.uint8, device=DEV); lab_a = torch.zeros(2 * p, Wn * p, dtype=torch.uint8, device=DEV)
    rgb_b, lab_b = torch.zeros_like(rgb_a), torch.zeros_like(lab_a)
    ops.argmax_stitch(logits.contiguous(), p, 1, Wn, pal, rgb_a, lab_a)
    ops.label_stitch(lab, p, 1, Wn, pal, rgb_b, lab_b)
    assert torch.equal(rgb_a, rgb_b) and torch.equal(lab_a, lab_b) and int(lab_a.sum()) > 0
