"""Whole-image inference with border correction and stitching, on the device.

Replaces the reference's per-tile loop (`utils.inference`, source/scripts/utils.py:145-171: batch-1 forward, CenterCrop,
CPU argmax, colourise, one PNG per tile) and the PNG re-stitching of source/scripts/inference.py:40-57: the full u8
image stays in HBM, tiles are gathered by index in batches, and argmax + palette lookup write straight into the
stitched map.  Border correction: the network sees an S x S window CENTRED on each p x p tile (S = border_correction).
"""
from __future__ import annotations

import torch

from . import ops
from .converters import GID15Converter


@torch.no_grad()
def segment_image(net, image_chw_u8: torch.Tensor, patch_size: int, border_correction: int | None = None, batch: int = 32,
                  tile_range=None, want_rgb: bool = True):
    """image u8 [3,H,W] on the GPU -> (rgb u8 [Hn*p, Wn*p, 3] | None, labels u8 [Hn*p, Wn*p]); Hn = H//p, Wn = W//p."""
    assert image_chw_u8.dtype == torch.uint8 and image_chw_u8.is_cuda and image_chw_u8.dim() == 3
    dev = image_chw_u8.device
    _, H, W = image_chw_u8.shape
    p = patch_size
    S = border_correction or p
    assert S >= p and (S - p) % 2 == 0 and S % 16 == 0, "border_correction must be >= patch_size, same parity, multiple of 16"
    Hn, Wn = H // p, W // p
    lo, hi = (0, Hn * Wn) if tile_range is None else tile_range
    labels = torch.zeros((Hn * p, Wn * p), dtype=torch.uint8, device=dev)
    rgb = torch.zeros((Hn * p, Wn * p, 3), dtype=torch.uint8, device=dev) if want_rgb else None
    palette = GID15Converter().palette_u8(dev)
    was_training = net.training
    net.eval()
    image = image_chw_u8.contiguous()
    for first in range(lo, hi, batch):
        n = min(batch, hi - first)
        tiles = torch.empty((n, 3, S, S), dtype=torch.uint8, device=dev)
        ops.crop_tiles(image, tiles, first, Wn, p)
        if hasattr(net, "predict_labels"):          # HIP network: fused head + argmax -> per-tile label maps
            lab = net.predict_labels(tiles)
        else:
            out = net(tiles, None)
            if getattr(net, "returns_logits", True):
                ops.argmax_stitch(out, p, first, Wn, palette, rgb, labels)
                continue
            lab = out.reshape(n, S, S).to(torch.uint8)   # a label-returning network (the Ensemble, S/utils.py:157-160)
        ops.label_stitch(lab.contiguous(), p, first, Wn, palette, rgb, labels)
    net.train(was_training)
    return rgb, labels
