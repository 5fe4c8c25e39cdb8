"""The reference's entry points (source/scripts/train.py, evaluation.py, inference.py) as kept at the repo root: run as
the user would run them, on synthetic data; plus train.py under torch.distributed.run with two ranks (rehearsed on the
one-GPU test box: both ranks on cuda:0, gloo standing in for RCCL)."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(tmp_path, **over):
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "synthetic_smoke.yaml")))
    cfg.update(checkpoint_directory=str(tmp_path), patch_size=64, batch_size=4, train="synthetic:2:8", validation="synthetic:1:8")
    cfg.update(over)
    path = os.path.join(tmp_path, "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    return path


def _run(cmd, env=None):
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    return r.stdout


@pytest.mark.parametrize("netname", ["Unetv2", "Resnet18Unet", "Resnet101", "TSwin", "SegformerMod", "MobileNet"])
def test_train_eval_inference_scripts(tmp_path, netname):
    """the three reference entry points end to end for the reference's own U-Net, BASELINE cfg 1's ResNet18-UNet, the reference's
    `Resnet101` (DeepLabV3, checkpoint in torchvision key names), `TSwin` (Swin-T + UPerNet) and the reference's `SegformerMod` and `MobileNet`"""
    from PIL import Image
    out = _run([sys.executable, "train.py", _cfg(tmp_path, epochs=2, net=netname)])
    assert "Training Done!" in out and "Saved checkpoint 2" in out and "mIoU" in out
    losses = [float(v) for v in re.findall(r"Reached (?:training|validation) loss: ([0-9.eE+-]+)", out)]
    assert len(losses) == 2 and all(np.isfinite(losses))
    ck = os.path.join(tmp_path, "checkpoint2")
    assert os.path.exists(ck)
    # evaluation.py on the checkpoint
    ecfg = os.path.join(tmp_path, "eval.yaml")
    yaml.safe_dump({"net": netname, "num_classes": 4, "device": "gpu", "precision": "bf16", "load_checkpoint": ck,
                    "dataset": "synthetic:1:8", "patch_size": 64, "ignore_background": True}, open(ecfg, "w"))
    out = _run([sys.executable, "evaluation.py", ecfg])
    assert "mIoU" in out
    # inference.py: one RGB image -> stitched colour map, with border correction
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 256, (200, 264, 3), dtype=np.uint8)).save(os.path.join(tmp_path, "img.png"))
    icfg = os.path.join(tmp_path, "inf.yaml")
    yaml.safe_dump({"net": netname, "num_classes": 4, "device": "gpu", "load_checkpoint": ck, "image": os.path.join(str(tmp_path), "img.png"),
                    "patch_size": 64, "border_correction": 96, "out_image": os.path.join(str(tmp_path), "out.png")}, open(icfg, "w"))
    _run([sys.executable, "inference.py", icfg])
    assert Image.open(os.path.join(tmp_path, "out.png")).size == (256, 192)     # (W // p) * p, (H // p) * p


@pytest.mark.parametrize("exact", [False, True])
def test_train_script_two_ranks(tmp_path, exact):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, CVCS_ONE_DEVICE="1", CVCS_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "train.py", _cfg(tmp_path, batch_size=2, exact_data_parallel=exact)]
    out = _run(cmd, env)
    assert out.count("Training Done!") == 2 and out.count("Saved checkpoint 1") == 1   # every rank trains, rank 0 saves
    assert "mIoU" in out


def _gid_dir(root):
    from PIL import Image
    rng = np.random.default_rng(0)
    for sub in ("Image__8bit_NirRGB", "Annotation__index", "Annotation__color"):
        os.makedirs(os.path.join(root, sub))
    for i in range(3):
        Image.fromarray(rng.integers(0, 256, (448, 672, 3), dtype=np.uint8)).save(os.path.join(root, "Image__8bit_NirRGB", f"im{i}.png"))
        Image.fromarray(rng.integers(0, 5, (448, 672), dtype=np.uint8)).save(os.path.join(root, "Annotation__index", f"im{i}.png"))
    return root


def test_device_tile_producer_matches_host_crops_bit_for_bit(tmp_path):
    """`Loader.get_iterable_chunk(device=...)` keeps the decoded images resident as u8 and produces every
    batch with one gather launch per source image - the tiles must equal the host crops (the reference's eager crop with
    v2.functional.crop's zero fill for shifted windows, S/dataset.py:25-32,136-172) bit for bit, for whole batches and for
    the per-rank shards; the class-weight scan (S/dataset.py:346-384) by the device histogram equals the host bincount."""
    import torch
    from cvcs_amd import dataset, utils
    root = _gid_dir(str(tmp_path / "gid"))
    mk = lambda dev: dataset.make_loader(root, 2, 224, 5, random_shift=True, seed=3, device=dev)   # noqa: E731
    host, devl = mk(None), mk("cuda:0")
    for ld in (host, devl):
        ld.shuffle()
    for c in range(len(host)):
        for shard in (None, (1, 2)):
            hb = list(utils.chunk_batches(host, c, 4, "cpu", shard=shard))
            db = list(utils.chunk_batches(devl, c, 4, "cuda:0", shard=shard))
            assert len(hb) == len(db) and len(hb) > 0
            for (hi, hm, _, _), (di, dm, _, _) in zip(hb, db):
                assert di.is_cuda and di.dtype == torch.uint8 and torch.equal(di.cpu(), hi) and torch.equal(dm.cpu(), hm)
    assert torch.equal(host.get_class_weights(5, True), devl.get_class_weights(5, True))


def test_train_script_on_an_image_directory(tmp_path):
    """train.py through the device-resident producer on a GID-15-shaped directory (224-pixel tiles, random shift, wCEL)"""
    root = _gid_dir(str(tmp_path / "gid"))
    os.makedirs(tmp_path / "ck")
    out = _run([sys.executable, "train.py", _cfg(tmp_path, checkpoint_directory=str(tmp_path / "ck"), train=root, validation=root,
                                                 patch_size=224, batch_size=3, chunk_size=2, validation_chunk_size=3, random_shift=True,
                                                 loss="wCEL", epochs=1)])
    assert "Training Done!" in out and "mIoU" in out
