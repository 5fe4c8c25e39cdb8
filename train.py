#!/usr/bin/env python3
"""`python3 train.py cfg.yaml` - the reference's training entry point (source/scripts/train.py) on the HIP path.

Same YAML keys (source/scripts/configs/train/README.txt); `train` / `validation` may be 'synthetic:<n_images>'.
Under `torch.distributed.run` every rank trains on its shard of each batch (cvcs_amd.parallel.DataParallel).
"""
import os
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
import sys
from pathlib import Path

import torch
import yaml

from cvcs_amd import dataset, utils

inFile = sys.argv[1]
with open(inFile, "r") as f:
    config = yaml.load(f, Loader=yaml.FullLoader)
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
if rank == 0:
    utils.display_configs(config)
NC = config["num_classes"] + 1
device = utils.load_device(config)
# the loaders draw every random decision (image order, tile order, +-20 px shift) from (seed, epoch, chunk): all ranks agree
Loader_train = dataset.make_loader(config["train"], config["chunk_size"], config["patch_size"], NC,
                                   random_shift=config.get("random_shift", False), seed=config.get("seed", 0), device=device)
Loader_validation = dataset.make_loader(config["validation"], config["validation_chunk_size"], config["patch_size"], NC, seed=1,
                                        device=device)
if config.get("debug"):
    Loader_train.specify([0, 1])
    Loader_validation.specify([0])

if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(device)
    backend = os.environ.get("CVCS_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device(device))
    else:
        dist.init_process_group(backend)
net = utils.load_network(config, device)
print("parameters", utils.count_params(net), "| tiles/epoch", len(Loader_train.idxs) * Loader_train.tpi, flush=True)
crit = utils.load_loss(config, device, Loader_train)
opt, scheduler = utils.load_optimizer(config, net)
if world > 1:
    from cvcs_amd.parallel import DataParallel
    net.flat_parameters()
    # exact_data_parallel: SyncBN + whole-batch loss mean, i.e. the reference's single-process step on world x batch tiles
    DataParallel(net, opt, exact=bool(config.get("exact_data_parallel", False)), criterion=crit)

training_loss_values, validation_loss_values, conf_flat = [], [], []
last_epoch = 0
if "load_checkpoint" in config:
    ck = torch.load(config["load_checkpoint"], map_location="cpu", weights_only=False)
    net.load_state_dict(ck["model_state_dict"])
    for obj, key in ((opt, "optimizer_state_dict"), (scheduler, "scheduler_state_dict")):
        try:
            obj.load_state_dict(ck[key])
            print(f"Restored {key}.")
        except Exception:
            print(f"{key} not compatible; {type(obj).__name__} starts fresh.")
    last_epoch = ck["epoch"] + 1
    Loader_train.epoch = last_epoch   # image order, tile order and shifts are functions of (seed, epoch): resume continues the sequence
    training_loss_values, validation_loss_values = ck["training_loss_values"], ck["validation_loss_values"]
assert Path(config["checkpoint_directory"]).is_dir(), "Please provide a valid directory to save checkpoints in."

for epoch in range(last_epoch, config["epochs"]):
    print("Started epoch {}".format(epoch + 1), flush=True)
    Loader_train.shuffle()
    for c in range(len(Loader_train)):
        # GPU: the chunk's images stay resident on the device, every batch is one gather launch;
        # N ranks: every rank takes its slice of the same global batch of batch_size * world tiles
        batches = utils.chunk_batches(Loader_train, c, config["batch_size"] * world, device, config.get("random_tps"),
                                      shard=(rank, world) if world > 1 else None)
        net.train()
        for image, index_mask, color_mask, context in batches:
            image, mask = image.to(device), utils.mask_reshape(index_mask.to(device))
            mask_pred = net(image, None)
            loss = crit(mask_pred, mask)
            training_loss_values.append(loss.item())
            opt.zero_grad()
            loss.backward()
            opt.step()
    if scheduler:
        scheduler.step()
    print("Running validation...", flush=True)
    validation_loss_values += utils.validation_loss(net, Loader_validation, crit, device, config["batch_size"])
    if (epoch + 1) % config["precision_evaluation_freq"] == 0:
        flat, normalized = utils.eval_model(net, Loader_validation, device, batch_size=config.get("eval_batch_size", 8),
                                            ignore_background=config["ignore_background"], num_classes=max(16, NC),
                                            shard=(rank, world) if world > 1 else None)
        if rank == 0:
            utils.print_metrics(flat.compute())
            conf_flat.append(flat.compute())
    if (epoch + 1) % config["freq"] == 0 and rank == 0:
        torch.save({"epoch": epoch, "model_state_dict": net.state_dict(), "optimizer_state_dict": opt.state_dict(),
                    "scheduler_state_dict": scheduler.state_dict(), "training_loss_values": training_loss_values,
                    "validation_loss_values": validation_loss_values, "batch_size": config["batch_size"],
                    "macro_precision": [], "weighted_precision": [], "conf_flat": conf_flat, "conf_normalized": [],
                    "optimizer": config["opt"]}, os.path.join(config["checkpoint_directory"], "checkpoint{}".format(epoch + 1)))
        print("Saved checkpoint {}".format(epoch + 1), flush=True)

print("Training Done!")
print(f"Reached training loss: {training_loss_values[-1]}")
print(f"Reached validation loss: {validation_loss_values[-1]}")
