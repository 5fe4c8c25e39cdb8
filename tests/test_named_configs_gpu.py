"""BASELINE.json's five configurations AS NAMED (model, tile size, class count), each against its CPU oracle at the full tile size:
eval-mode logits of the f32 path within 1e-3 of max|logit|, argmax mismatches counted and each a near-tie; the bf16 path's labels
agree with the oracle's on all but a counted handful of near-tie pixels; one full bf16 train step at the named size is finite and
bitwise reproducible.  (Batch sizes are cut to what the CPU oracle finishes in seconds; the per-tile work is the named one.  The
multi-GPU part of configs 3-5 is covered by tests/test_dataparallel_gpu.py and the driver's scaling run; the "mixed bf16/fp8 convs" of
config 5 as named - 1024 x 1024 tiles, 20 classes - are checked layer by layer against the fp8 emulation in
tests/test_fp8_gpu.py::test_fp8_network_layer_by_layer_forward_and_backward.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from cvcs_amd import nets, utils  # noqa: E402
from oracle import deeplab_oracle as D  # noqa: E402
from oracle import resnet_unet_oracle as R  # noqa: E402
from oracle import swin_upernet_oracle as W  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402

DEV = "cuda:0"

CONFIGS = {
    # name: (network class, oracle init, oracle forward, classes incl. background, batch, tile)
    "cfg1 ResNet18-UNet 4 classes 8x256x256": (lambda nc, p: nets.Resnet18Unet(nc, p), lambda nc: R.init_params("resnet18", nc, seed=3),
                                               lambda p, x: R.forward(p, x, "resnet18", train=False), 5, 8, 256),
    "cfg2 ResNet50-UNet 512x512": (lambda nc, p: nets.Resnet50Unet(nc, p), lambda nc: R.init_params("resnet50", nc, seed=3),
                                   lambda p, x: R.forward(p, x, "resnet50", train=False), 16, 2, 512),
    "cfg3 DeepLabV3+ ResNet50 ASPP 6/12/18 512x512": (lambda nc, p: nets.DeepLabV3Plus(nc, p), lambda nc: D.init_params("resnet50", nc, seed=3, plus=True),
                                                      lambda p, x: D.forward(p, x, "resnet50", train=False, output_stride=16, plus=True), 16, 2, 512),
    "cfg4 Swin-T + UPerNet 512x512 window 7": (lambda nc, p: nets.SwinTUperNet(nc, p), lambda nc: W.init_params(nc, seed=3),
                                               lambda p, x: W.forward(p, x, train=False), 16, 2, 512),
    "cfg5 ResNet50-UNet 1024x1024 20 classes": (lambda nc, p: nets.Resnet50Unet(nc, p), lambda nc: R.init_params("resnet50", nc, seed=3),
                                                lambda p, x: R.forward(p, x, "resnet50", train=False), 21, 1, 1024),
}


def _net(make, init, NC, precision):
    net = make(NC, precision)
    missing, unexpected = net.load_state_dict(init(NC), strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    return net.to(DEV)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_named_config_eval_logits_and_labels_match_the_cpu_oracle(name):
    make, init, fwd, NC, B, S = CONFIGS[name]
    img, _ = O.synthetic_tiles(B, S, NC, seed=11, structured=True)
    with torch.no_grad():
        want = fwd(init(NC), img.float())
    scale = want.abs().max().item()
    top2 = want.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    ref_lab = want.argmax(1)
    for precision, tol in (("fp32", 1e-3), ("bf16", None)):
        net = _net(make, init, NC, precision)
        net.eval()
        with torch.no_grad():
            got = net(img.to(DEV), None).cpu()
            lab = net.predict_labels(img.to(DEV)).cpu().long()
        err = (got - want).abs().max().item() / scale
        rms = (got - want).pow(2).mean().sqrt().item() / scale
        bad = lab != ref_lab
        nbad = int(bad.sum())
        worst_margin = margin[bad].max().item() / scale if nbad else 0.0
        print(f"{name} [{precision}]: logits max {err:.2e} rms {rms:.2e} of max|logit|; {nbad} of {lab.numel()} labels differ, "
              f"largest oracle top-2 margin among them {worst_margin:.2e}")
        assert torch.equal(lab, got.argmax(1))           # the fused head + argmax launch agrees with the logits it would have written
        if tol is not None:
            assert err <= tol, (name, err)
            assert worst_margin <= 2 * tol               # every differing label is a near-tie of the oracle itself
            assert nbad <= 1e-4 * lab.numel()
        else:
            assert rms <= 1e-2 and worst_margin <= 4 * max(err, 1e-3)
            assert nbad <= 3e-2 * lab.numel()      # (measured 0.1 - 2.3 %: random-init networks, a 16-channel last stage, tiny class margins)


@pytest.mark.parametrize("name", ["cfg4 Swin-T + UPerNet 512x512 window 7", "cfg5 ResNet50-UNet 1024x1024 20 classes"])
def test_named_config_bf16_train_step_is_finite_and_reproducible(name):
    make, init, _, NC, _, S = CONFIGS[name]
    B = 2
    img, lab = O.synthetic_tiles(B, S, NC, seed=12, structured=True)
    runs = []
    for _ in range(2):
        net = _net(make, init, NC, "bf16")
        crit = utils.CrossEntropyLoss(ignore_index=0)
        optim, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 20}, net)
        net.train()
        losses = []
        for _ in range(3):
            loss = crit(net(img.to(DEV), None), lab.to(DEV))
            optim.zero_grad()
            loss.backward()
            optim.step()
            losses.append(loss.item())
        flat = net.flat_parameters()[0]
        assert torch.isfinite(flat).all() and all(map(lambda v: v == v, losses))
        runs.append((losses, flat.clone()))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    assert runs[0][0][-1] < runs[0][0][0]
