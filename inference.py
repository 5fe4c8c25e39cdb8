#!/usr/bin/env python3
"""`python3 inference.py cfg.yaml` - the reference's inference entry point (source/scripts/inference.py:12-57) on the
HIP path.  Keys: dataset (dir with Image__8bit_NirRGB) | image (one file), net, load_checkpoint, patch_size,
border_correction, range [lo, hi] (tile indices), out_image, num_classes, device; optional batch_size."""
import os
import sys

import numpy as np
import torch
import yaml
from PIL import Image

from cvcs_amd import utils
from cvcs_amd.inference import segment_image

with open(sys.argv[1], "r") as f:
    config = yaml.load(f, Loader=yaml.FullLoader)
device = utils.load_device(config)
net = utils.load_network(config, device)
if "load_checkpoint" in config:
    utils.load_checkpoint(config, net)
if "image" in config:
    path = config["image"]
else:
    idir = os.path.join(config["dataset"], "Image__8bit_NirRGB")
    path = os.path.join(idir, sorted(os.listdir(idir))[config.get("image_index", 0)])
img = torch.from_numpy(np.array(Image.open(path).convert("RGB"))).permute(2, 0, 1).contiguous().to(device)
rng = tuple(config["range"]) if config.get("range") else None
rgb, labels = segment_image(net, img, config["patch_size"], config.get("border_correction"), config.get("batch_size", 32), rng)
Image.fromarray(rgb.cpu().numpy()).save(config.get("out_image", "out.png"))
print("wrote", config.get("out_image", "out.png"), tuple(rgb.shape))
