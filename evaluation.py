#!/usr/bin/env python3
"""`python3 evaluation.py cfg.yaml` - the reference's evaluation entry point (source/scripts/evaluation.py:8-33)."""
import sys

import torch
import yaml

from cvcs_amd import dataset, utils

with open(sys.argv[1], "r") as f:
    config = yaml.load(f, Loader=yaml.FullLoader)
device = utils.load_device(config)
net = utils.load_network(config, device)
if "load_checkpoint" in config:
    ck = torch.load(config["load_checkpoint"], map_location="cpu", weights_only=False)
    net.load_state_dict(ck["model_state_dict"])
NC = config["num_classes"] + 1
loader = dataset.make_loader(config["dataset"], 1, config["patch_size"], NC, seed=1)
if config.get("images"):
    loader.specify(config["images"])
flat, _ = utils.eval_model(net, loader, device, batch_size=config.get("eval_batch_size", 8),
                           ignore_background=config.get("ignore_background", False), num_classes=max(16, NC))
utils.print_metrics(flat.compute())
