"""N > 1 on the device: two ranks (two processes sharing cuda:0, collectives over gloo - RCCL wants one GPU per rank,
and the test box has one) against ONE process that trains on the whole batch.

With `DataParallel(exact=True)` (SyncBN moments + whole-batch loss normalisation) the
sharded run must reproduce the reference's single-process step on the global batch up to summation order."""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

NC, B, S, STEPS = 5, 4, 64, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _train(img, lab, wrap, weights):
    from cvcs_amd import utils
    dev = "cuda:0"
    torch.manual_seed(0)
    net = utils.load_network({"net": "Unetv2", "num_classes": NC - 1, "precision": "fp32"}, dev)
    crit = utils.CrossEntropyLoss(weight=weights, ignore_index=0)
    optim, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 1}, net)
    net.train()
    if wrap is not None:
        net.flat_parameters()
        wrap(net, optim, crit)
    losses = []
    for _ in range(STEPS):
        loss = crit(net(img.to(dev), None), lab.to(dev))
        optim.zero_grad(); loss.backward(); optim.step()
        losses.append(loss.item())
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    return losses, sd


def _confusion(shard):
    """confusion matrix of a freshly seeded (untrained) network over a synthetic validation loader"""
    from cvcs_amd import dataset, utils
    torch.manual_seed(3)
    net = utils.load_network({"net": "Unetv2", "num_classes": NC - 1, "precision": "fp32"}, "cuda:0")
    loader = dataset.make_loader("synthetic:3:4", 2, S, NC, seed=1)
    flat, _ = utils.eval_model(net, loader, "cuda:0", batch_size=2, ignore_background=True, num_classes=16, shard=shard)
    return flat.compute()


def _worker(rank, world, port, path):
    import torch.distributed as dist
    from oracle import unet_oracle as O
    from cvcs_amd.parallel import DataParallel, shard_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        img, lab = O.synthetic_tiles(B, S, NC, seed=21, structured=True)
        lo, hi = shard_batch(B, rank, world)
        w = torch.tensor([0.0, 0.7, 1.3, 0.9, 1.1])
        losses, sd = _train(img[lo:hi], lab[lo:hi], lambda n, o, c: DataParallel(n, o, bucket_mb=8.0, exact=True, criterion=c), w)
        conf = _confusion((rank, world))
        torch.save({"losses": losses, "sd": sd, "conf": conf}, f"{path}/rank{rank}.pt")
    finally:
        dist.destroy_process_group()


def test_two_ranks_exact_mode_match_one_process_on_the_whole_batch():
    from oracle import unet_oracle as O
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)
        r0, r1 = (torch.load(f"{tmp}/rank{r}.pt") for r in range(2))
    img, lab = O.synthetic_tiles(B, S, NC, seed=21, structured=True)
    w = torch.tensor([0.0, 0.7, 1.3, 0.9, 1.1])
    losses, sd = _train(img, lab, None, w)
    # sharded evaluation: the summed count matrix is the single-process one, bit for bit
    conf = _confusion(None)
    assert conf.sum() > 0 and torch.equal(r0["conf"], conf) and torch.equal(r1["conf"], conf)
    # both ranks hold the same model, running statistics and (whole-batch) loss values
    assert r0["losses"] == pytest.approx(r1["losses"], rel=1e-6)
    for k in sd:
        assert torch.allclose(r0["sd"][k].float(), r1["sd"][k].float(), rtol=1e-6, atol=1e-7), k
    # ... and they are the single-process run on all B tiles, up to fp32 summation order
    # (the third loss sits behind two optimiser steps, which amplify the last-bit differences of the first)
    assert r0["losses"] == pytest.approx(losses, rel=1e-4)
    rows = []
    for k, v in sd.items():
        if v.dtype != torch.float32:
            continue
        diff = r0["sd"][k] - v
        rows.append((diff.norm().item() / (v.norm().item() + 1e-12), diff.abs().max().item(), k))
    rows.sort(reverse=True)
    for r in rows[:8]:
        print("rel %.2e  absmax %.2e  %s" % r)
    # BN biases start at 0 and move by lr * gradient only, so their relative figure is that of a GRADIENT (ReLU / pooling
    # decisions that flip under a different summation order move it by ~1e-3); everything is tiny in absolute terms
    assert all(rel < 2e-4 or absmax < 2e-5 for rel, absmax, _ in rows), rows[:4]


# ---------------------------------------------------------------------------------------------------- ResNet-UNet (replayed launch plans)
def _resnet_net(seed=0, name="Resnet18Unet"):
    from cvcs_amd import utils
    torch.manual_seed(seed)
    net = utils.load_network({"net": name, "num_classes": NC - 1, "precision": "fp32"}, "cuda:0")
    crit = utils.CrossEntropyLoss(ignore_index=0)
    optim, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 1}, net)
    net.train()
    return net, crit, optim


def _resnet_worker(rank, world, port, path, name="Resnet18Unet"):
    import torch.distributed as dist
    from oracle import unet_oracle as O
    from cvcs_amd.parallel import DataParallel, shard_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        img, lab = O.synthetic_tiles(B, S, NC, seed=22, structured=True)
        lo, hi = shard_batch(B, rank, world)
        net, crit, optim = _resnet_net(name=name)
        net.flat_parameters()
        DataParallel(net, optim, bucket_mb=4.0)      # torch-DDP semantics: per-rank BatchNorm statistics, averaged gradients
        flats, losses = [], []
        for _ in range(STEPS):
            loss = crit(net(img[lo:hi].to("cuda:0"), None), lab[lo:hi].to("cuda:0"))
            optim.zero_grad(); loss.backward(); optim.step()
            torch.cuda.synchronize()
            flats.append(net.flat_parameters()[0].detach().cpu().clone())
            losses.append(loss.item())
        torch.save({"flats": flats, "losses": losses}, f"{path}/rank{rank}.pt")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["Resnet18Unet", "TSwin", "SegformerMod", "MobileNet"])
def test_resnet_unet_two_ranks_average_their_gradients_through_the_replayed_plan(name):
    """(also Swin-T + UPerNet, whose flat parameter order is arranged for exactly this: nets.SwinTUperNet._flat_order)
    the bucket hooks are host callbacks INSIDE the recorded backward launch list (resnet_engine._ready): two ranks must
    hold bit-identical parameters after every step, and the first update must be the fused SGD2 step on the MEAN of the two
    shards' gradients (each computed here by a plain single-process backward on that shard)"""
    from oracle import unet_oracle as O
    from cvcs_amd.parallel import shard_batch
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_resnet_worker, args=(2, _free_port(), tmp, name), nprocs=2, join=True)
        r0, r1 = (torch.load(f"{tmp}/rank{r}.pt") for r in range(2))
    for a, b in zip(r0["flats"], r1["flats"]):
        assert torch.equal(a, b)
    img, lab = O.synthetic_tiles(B, S, NC, seed=22, structured=True)
    grads = []
    for r in range(2):
        lo, hi = shard_batch(B, r, 2)
        net, crit, optim = _resnet_net(name=name)
        crit(net(img[lo:hi].to("cuda:0"), None), lab[lo:hi].to("cuda:0")).backward()
        torch.cuda.synchronize()
        grads.append(net.flat_parameters()[1].detach().clone())
    net, crit, optim = _resnet_net(name=name)
    flat, flat_grad = net.flat_parameters()
    flat_grad.copy_((grads[0] + grads[1]) / 2)
    optim.step()
    torch.cuda.synchronize()
    want = flat.detach().cpu()
    diff = (r0["flats"][0] - want).abs().max().item()
    assert diff <= 1e-6 * want.abs().max().item(), diff
    assert r0["losses"][-1] < r0["losses"][0]


# ---------------------------------------------------------------------------------------------------- exact mode on the launch-plan engines
def _exact_train(name, img, lab, wrap):
    from cvcs_amd import utils
    torch.manual_seed(0)
    # (aspp_dropout 0: the counter-based Dropout mask of `MobileNet` is a function of the LOCAL element index - a sharded run draws another mask)
    net = utils.load_network({"net": name, "num_classes": NC - 1, "precision": "fp32", "aspp_dropout": 0.0}, "cuda:0")
    crit = utils.CrossEntropyLoss(ignore_index=0)
    optim, _ = utils.load_optimizer({"opt": "SGD2", "epochs": 1}, net)
    net.train()
    if wrap is not None:
        net.flat_parameters()
        wrap(net, optim, crit)
    losses = []
    for _ in range(STEPS):
        loss = crit(net(img.to("cuda:0"), None), lab.to("cuda:0"))
        optim.zero_grad(); loss.backward(); optim.step()
        losses.append(loss.item())
    return losses, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}


S_EXACT = 128      # (at 64 x 64 the last ResNet stage is a 2 x 2 map: its BatchNorms average 16 values and amplify summation-order noise to percents)


def _exact_worker(rank, world, port, path, name):
    import torch.distributed as dist
    from oracle import unet_oracle as O
    from cvcs_amd.parallel import DataParallel, shard_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        img, lab = O.synthetic_tiles(B, S_EXACT, NC, seed=23, structured=True)
        lo, hi = shard_batch(B, rank, world)
        losses, sd = _exact_train(name, img[lo:hi], lab[lo:hi], lambda n, o, c: DataParallel(n, o, bucket_mb=8.0, exact=True, criterion=c))
        torch.save({"losses": losses, "sd": sd}, f"{path}/rank{rank}.pt")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["Resnet50Unet", "DeepLabV3Plus", "TSwin", "MobileNet"])
def test_exact_mode_on_the_launch_plan_engines_two_ranks_equal_one_process_on_the_whole_batch(name):
    """`DataParallel(exact=True)` for the networks BASELINE's data-parallel configurations name (ResNet50-UNet cfg 5, DeepLabV3+ cfg 3,
    Swin-T + UPerNet cfg 4): every BatchNorm's batch moments and backward sums are exchanged from host callbacks recorded inside the
    replayed launch plans (resnet_engine.enable_sync_bn), the loss mean spans both ranks' pixels - two ranks x 2 tiles must be one process
    on 4 tiles up to summation order (three SGD2 steps, f32)."""
    from oracle import unet_oracle as O
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_exact_worker, args=(2, _free_port(), tmp, name), nprocs=2, join=True)
        r0, r1 = (torch.load(f"{tmp}/rank{r}.pt") for r in range(2))
    img, lab = O.synthetic_tiles(B, S_EXACT, NC, seed=23, structured=True)
    losses, sd = _exact_train(name, img, lab, None)
    assert r0["losses"] == pytest.approx(r1["losses"], rel=1e-6)
    for k in sd:
        assert torch.allclose(r0["sd"][k].float(), r1["sd"][k].float(), rtol=1e-6, atol=1e-7), k
    # Yardstick for "up to summation order": the SAME single-process run with the four tiles in reverse order - mathematically the same
    # step (BatchNorm and the loss mean are permutation-invariant), a different floating-point summation order.  Three SGD2 steps of a
    # 50-layer network with BatchNorms over 64-value maps turn that into percent-level differences of the deepest running means and
    # of near-zero BatchNorm biases; the sharded run may not be further from the single-process one than a few times that.
    losses_perm, sd_perm = _exact_train(name, img.flip(0), lab.flip(0), None)
    print(f"{name} losses: sharded {r0['losses']} | single {losses} | single, tiles reversed {losses_perm}")
    assert r0["losses"][0] == pytest.approx(losses[0], rel=2e-5)                 # the first step is a pure function of the inputs
    # (MobileNet: 60 BatchNorms, ReLU / hardswish kinks on 8 x 8 maps - ONE sample of the permutation noise is a poor yardstick for another
    #  summation order's; measured: steps 1-2 agree to 1e-7, step 3 to 7e-5 against a permutation distance of 1e-5: 16 x and 2e-4 there)
    mult, floor = (16.0, 2e-4) if name == "MobileNet" else (4.0, 2e-5)
    for a, b, c in zip(r0["losses"], losses, losses_perm):
        assert abs(a - b) <= mult * abs(c - b) + floor * abs(b), (r0["losses"], losses, losses_perm)
    worst = (0.0, "")
    for k, v in sd.items():
        if v.dtype != torch.float32:
            continue
        noise = (sd_perm[k] - v).abs().max().item()
        diff = (r0["sd"][k] - v).abs().max().item()
        rel = diff / (v.abs().max().item() + 1e-12)
        ok = diff <= mult * noise + 1e-6 or rel < 10 * floor
        worst = max(worst, (diff / (noise + 1e-12) if not rel < 10 * floor else 0.0, k))
        assert ok, (k, diff, noise, rel)
    print(f"{name}: largest (sharded - single) / (permuted - single) parameter difference ratio {worst[0]:.2f} ({worst[1]})")
