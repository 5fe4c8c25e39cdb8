"""Tile producers with the reference's `Loader` API (source/scripts/dataset.py:228-387): `len(loader)` chunks,
`get_iterable_chunk(c, random_tps=None)`, `shuffle()`, `specify(idxs)`, `patch_size`, `tpi`, `idxs`,
`get_class_weights(classes, ignore_background)`.

`Loader` reads GID-15 directories (Image__8bit_NirRGB / Annotation__index / Annotation__color) with PIL + numpy
(torchvision is not a dependency here); `SyntheticLoader` produces the structured synthetic tiles of SURVEY
section 8(d) (Voronoi label maps, per-class colour + noise) for parity / mIoU runs without imagery.
The per-tile hot path consumes uint8 tensors: [3,p,p] image and [p,p] index mask.
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch


class _Chunk(torch.utils.data.IterableDataset):
    def __init__(self, patches):
        super().__init__()
        self.patches = patches
        self.chunk_crops = patches

    def __iter__(self):
        return iter(self.patches)

    def __len__(self):
        return len(self.patches)


def class_weights_from_counts(counts: torch.Tensor, ignore_background: bool):
    """S/dataset.py:360-384: w_j = N_tot / (bins * N_j); 0 for empty classes; background excluded on request."""
    c = counts[1:] if ignore_background else counts
    tot, bins = torch.sum(c), len(c)
    w = torch.tensor([0.0 if n.item() == 0 else (tot / (bins * n.item())).item() for n in c], dtype=torch.float32)
    return torch.concat((torch.tensor([0.0]), w)) if ignore_background else w


class SyntheticLoader:
    """`n_images` pseudo-images of `tiles_per_image` tiles each; deterministic in (seed, image index)."""

    def __init__(self, n_images, chunk_size, patch_size=256, num_classes=16, tiles_per_image=16, seed=0):
        self.patch_size, self.NC, self.tpi = patch_size, num_classes, tiles_per_image
        self.chunk_size, self.seed = chunk_size, seed
        self.idxs = list(range(n_images))
        self._base_idxs = list(self.idxs)
        self.epoch = 0
        self._gen_chunks()
        g = torch.Generator().manual_seed(99)
        self.base = torch.randint(30, 226, (num_classes, 3), generator=g).float()
        self.count = None

    def _gen_chunks(self):
        self.chunks = [self.idxs[i:i + self.chunk_size] for i in range(0, len(self.idxs), self.chunk_size)]

    def __len__(self):
        return len(self.chunks)

    def shuffle(self):
        """image order of the next epoch: a function of (seed, epoch) only, so that every rank of a data-parallel job draws
        the SAME permutation (the reference shuffles with the process-global `random`, S/dataset.py:289-291, which is only
        right for its single process)"""
        self.epoch += 1
        self.idxs = list(self._base_idxs)      # a PURE function of (seed, epoch): a run resumed at epoch k (train.py sets
        random.Random(self.seed * 1000003 + self.epoch).shuffle(self.idxs)   # .epoch) draws the orders of epochs k+1.. again
        self._gen_chunks()

    def specify(self, targets):
        self.idxs = [self.idxs[i] for i in targets]
        self._base_idxs = list(self.idxs)
        self._gen_chunks()

    def _tile(self, g):
        S, NC = self.patch_size, self.NC
        yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
        n = int(torch.randint(8, 33, (1,), generator=g))
        sy, sx = torch.randint(0, S, (n,), generator=g), torch.randint(0, S, (n,), generator=g)
        cls = torch.randint(1, NC, (n,), generator=g)
        cls = torch.where(torch.rand(n, generator=g) < 0.05, torch.zeros_like(cls), cls)
        d = (yy[None] - sy[:, None, None]) ** 2 + (xx[None] - sx[:, None, None]) ** 2
        lab = cls[d.argmin(0)]
        img = self.base[lab].permute(2, 0, 1) + 20.0 * torch.randn(3, S, S, generator=g)
        return img.clamp(0, 255).to(torch.uint8), lab.to(torch.uint8)

    def get_iterable_chunk(self, idx, random_tps=None, device=None, batch_size=None, shard=None):
        """device=None: the reference's contract - an iterable of per-tile tuples for a DataLoader.  device given: the tiles
        of the chunk are uploaded once and the iterable yields whole batches resident on the device (see DeviceTiles)."""
        if random_tps:
            raise ValueError("random_tps (random rescaled crops of the full images, S/dataset.py:173-198) needs a dataset of full images: "
                             "the synthetic loader generates tiles, not images")
        patches = []
        for image in self.chunks[idx]:
            g = torch.Generator().manual_seed(self.seed * 100003 + image)
            for _ in range(self.tpi):
                img, lab = self._tile(g)
                patches.append((img, lab, torch.zeros(1), torch.zeros(1)))
        if device is not None and torch.device(device).type == "cuda":
            return DeviceTiles(torch.stack([p[0] for p in patches]).to(device), torch.stack([p[1] for p in patches]).to(device),
                               batch_size or 1, shard)
        return _Chunk(patches)

    def get_class_weights(self, classes, ignore_background=False):
        if self.count is None:
            self.count = torch.zeros(classes)
            for c in range(len(self)):
                for _, lab, _, _ in self.get_iterable_chunk(c):
                    self.count += torch.bincount(lab.reshape(-1).long(), minlength=classes)[:classes].float()
        return class_weights_from_counts(self.count, ignore_background)


class Loader:
    """GID-15 chunk loader (S/dataset.py:228-387) without torchvision: full images decoded with PIL, every
    non-overlapping `patch_size` tile of a chunk's images cropped eagerly (optionally shifted by +-20 px)."""

    def __init__(self, root, chunk_size, random_shift=False, patch_size=224, image_transforms=None, mask_transforms=None,
                 load_context=False, load_color_mask=False, seed=0):
        from PIL import Image  # noqa: F401
        assert patch_size in (224, 256, 512), "patch_size must be 224, 256 or 512"   # S/dataset.py:268
        self.root, self.chunk_size, self.random_shift, self.patch_size = root, chunk_size, random_shift, patch_size
        idir, mdir = os.path.join(root, "Image__8bit_NirRGB"), os.path.join(root, "Annotation__index")
        self.images = sorted(os.path.join(idir, f) for f in os.listdir(idir))
        self.index_masks = sorted(os.path.join(mdir, f) for f in os.listdir(mdir))
        assert len(self.images) == len(self.index_masks)
        from PIL import Image as I
        with I.open(self.images[0]) as im:
            W, H = im.size
        self.tpi = (H // patch_size) * (W // patch_size)
        self.H, self.W = H, W
        self.idxs = list(range(len(self.images)))
        self._base_idxs = list(self.idxs)
        self.seed, self.epoch = seed, 0
        self.device = None     # set (make_loader(device=...)) to count the class pixels with the device histogram
        self._gen_chunks()
        self.count = None

    def _gen_chunks(self):
        self.chunks = [self.idxs[i:i + self.chunk_size] for i in range(0, len(self.idxs), self.chunk_size)]

    def __len__(self):
        return len(self.chunks)

    def shuffle(self):
        """a function of (seed, epoch): identical on every rank of a data-parallel job (see SyntheticLoader.shuffle)"""
        self.epoch += 1
        self.idxs = list(self._base_idxs)      # a PURE function of (seed, epoch): a run resumed at epoch k (train.py sets
        random.Random(self.seed * 1000003 + self.epoch).shuffle(self.idxs)   # .epoch) draws the orders of epochs k+1.. again
        self._gen_chunks()

    def specify(self, targets):
        self.idxs = [self.idxs[i] for i in targets]
        self._base_idxs = list(self.idxs)
        self._gen_chunks()

    def _decode(self, i):
        from PIL import Image
        img = torch.from_numpy(np.array(Image.open(self.images[i]).convert("RGB"))).permute(2, 0, 1).contiguous()
        return img, self._decode_mask(i)

    def _decode_mask(self, i):
        """the index mask alone (the class-weight scan, S/dataset.py:346-384, never needs the RGB image)"""
        from PIL import Image
        mask = torch.from_numpy(np.array(Image.open(self.index_masks[i])))
        if mask.dim() == 3:
            mask = mask[..., 0]
        return mask.to(torch.uint8).contiguous()

    def plan_items(self, idx):
        """(position in the chunk, top, left) of every tile of chunk idx in the order they are served: the +-20 px random
        shift (S/dataset.py:25-26,143) and the tile shuffle (S/dataset.py:198) are drawn from a generator seeded with
        (seed, epoch, chunk) - the host path, the device path and every rank produce the same list."""
        p = self.patch_size
        rng = random.Random((self.seed * 1000003 + self.epoch) * 1000003 + idx)
        items = []
        for pos in range(len(self.chunks[idx])):
            for ty in range(self.H // p):
                for tx in range(self.W // p):
                    y, x = ty * p, tx * p
                    if self.random_shift:
                        y, x = y + rng.randint(-20, 20), x + rng.randint(-20, 20)
                    items.append((pos, y, x))
        rng.shuffle(items)
        return items

    def get_iterable_chunk(self, idx, random_tps=None, device=None, batch_size=None, shard=None):
        """device=None: the reference's contract (S/dataset.py:331-335) - an iterable of per-tile tuples for a DataLoader,
        cropped on the host.  device given: the chunk's images are decoded once, kept resident on the device as u8, and
        the iterable yields whole batches produced by one gather launch per source image (DeviceChunk)."""
        decoded = [self._decode(i) for i in self.chunks[idx]]
        items = self.plan_items(idx)
        p = self.patch_size
        # random_tps = [(aug_size, percentage), ...] (S/dataset.py:173-198, read at S/train.py:109): extra random rescaled crops, appended and
        # shuffled in; planned from the chunk's own generator, so every rank and both paths below serve the same tiles
        rng = random.Random(((self.seed * 1000003 + self.epoch) * 1000003 + idx) * 31 + 7)
        extra = plan_random_tps(random_tps, len(items), len(decoded), self.H, self.W, rng)
        if device is not None and torch.device(device).type == "cuda":
            images, masks = [im.to(device) for im, _ in decoded], [m.to(device) for _, m in decoded]
            if extra:
                # the rescaled crops become one more resident "image": a strip of p x p tiles the gather launch reads like any other source
                tiles = [resize_tile(images[pos][:, y:y + a, x:x + a], masks[pos][y:y + a, x:x + a], p) for pos, y, x, a in extra]
                images.append(torch.cat([t[0] for t in tiles], dim=1).contiguous())
                masks.append(torch.cat([t[1] for t in tiles], dim=0).contiguous())
                items = items + [(len(decoded), j * p, 0) for j in range(len(extra))]
                rng.shuffle(items)
            return DeviceChunk(images, masks, p, batch_size or 1, items=items, shard=shard)
        patches = [(crop_zero_filled(decoded[pos][0], y, x, p), crop_zero_filled(decoded[pos][1], y, x, p), torch.zeros(1), torch.zeros(1))
                   for pos, y, x in items]
        if extra:
            for pos, y, x, a in extra:
                im, m = resize_tile(decoded[pos][0][:, y:y + a, x:x + a], decoded[pos][1][y:y + a, x:x + a], p)
                patches.append((im, m, torch.zeros(1), torch.zeros(1)))
            order = list(range(len(patches)))
            rng.shuffle(order)
            patches = [patches[j] for j in order]
        return _Chunk(patches)

    def get_class_weights(self, classes, ignore_background=False):
        if self.count is None:
            if self.device is not None and torch.device(self.device).type == "cuda":
                from . import ops
                counts = torch.zeros(classes, dtype=torch.int64, device=self.device)
                for i in range(len(self.index_masks)):
                    ops.label_histogram(self._decode_mask(i).to(self.device), counts)
                self.count = counts.cpu().float()
            else:
                self.count = torch.zeros(classes, dtype=torch.float32)
                for i in range(len(self.index_masks)):
                    m = self._decode_mask(i).reshape(-1).long()
                    self.count += torch.bincount(m, minlength=classes)[:classes].float()
        return class_weights_from_counts(self.count, ignore_background)


def resize_tile(img: torch.Tensor, mask: torch.Tensor, p: int):
    """the reference's two resizers on a random rescaled crop (S/dataset.py:129-130,192-195): image `v2.Resize(p, BILINEAR)` (antialiased,
    rounded back to u8), index mask `v2.Resize(p, NEAREST_EXACT)`.  img u8 [3,a,a], mask u8 [a,a] (host or device) -> u8 [3,p,p], u8 [p,p]"""
    f = torch.nn.functional.interpolate(img[None].float(), size=(p, p), mode="bilinear", align_corners=False, antialias=True)[0]
    m = torch.nn.functional.interpolate(mask[None, None].float(), size=(p, p), mode="nearest-exact")[0, 0]
    return f.round().clamp(0, 255).to(torch.uint8), m.to(torch.uint8)


def plan_random_tps(random_tps, n_tiles, n_images, H, W, rng):
    """S/dataset.py:173-198: for every (aug_size, percentage) add int(percentage * n_tiles) random aug_size x aug_size crops of random images
    of the chunk (each later resized to the patch size).  The draws come from `rng` (seeded with (seed, epoch, chunk)): every rank and both the
    host and the device path plan the same crops.  -> [(position in the chunk, top, left, aug_size)]"""
    extra = []
    for aug_size, percentage in random_tps or ():
        aug_size = int(aug_size)
        if not 0 < aug_size < min(H, W):
            raise ValueError(f"random_tps: crop size {aug_size} does not fit a {H} x {W} image")
        for _ in range(int(percentage * n_tiles)):
            extra.append((rng.randint(0, n_images - 1), rng.randint(0, H - 1 - aug_size), rng.randint(0, W - 1 - aug_size), aug_size))
    return extra


def crop_zero_filled(t: torch.Tensor, y: int, x: int, p: int):
    """t[..., y:y+p, x:x+p] with the part of the window that lies outside the tensor filled with 0 - the semantics of
    torchvision's v2.functional.crop that the reference's shifted tiles get (S/dataset.py:25-32,143): image pixels 0, index
    mask 0 = the ignored class."""
    H, W = t.shape[-2:]
    out = torch.zeros(t.shape[:-2] + (p, p), dtype=t.dtype)
    y0, y1, x0, x1 = max(y, 0), min(y + p, H), max(x, 0), min(x + p, W)
    if y1 > y0 and x1 > x0:
        out[..., y0 - y:y1 - y, x0 - x:x1 - x] = t[..., y0:y1, x0:x1]
    return out


class DeviceTiles:
    """tiles already resident on the device as [N,3,p,p] / [N,p,p] u8: yields batches (or this rank's shard of them)"""

    def __init__(self, images, masks, batch_size, shard=None):
        self.images, self.masks, self.bs, self.shard = images, masks, batch_size, shard
        self.chunk_crops = range(images.shape[0])

    def __len__(self):
        n = self.images.shape[0]
        return n // self.bs if self.shard else (n + self.bs - 1) // self.bs

    def __iter__(self):
        n = self.images.shape[0]
        for b0 in range(0, n, self.bs):
            lo, hi = b0, min(b0 + self.bs, n)
            if self.shard:
                if hi - lo < self.bs:
                    return          # drop the ragged last global batch (every rank must take the same number of steps)
                r, w = self.shard
                per = self.bs // w
                lo, hi = b0 + r * per, b0 + (r + 1) * per
            yield self.images[lo:hi], self.masks[lo:hi], torch.zeros(1), torch.zeros(1)


class DeviceChunk:
    """A chunk whose full images (u8 CHW) and index masks (u8 HW) are RESIDENT on the device; batches are produced by one
    gather launch each (cvcs_gather_tiles) instead of the reference's eager per-patch Python crop (S/dataset.py:136-172).
    Iterating yields (image u8 [B,3,p,p], index_mask u8 [B,p,p], color_mask, context) already on the device, i.e. what
    the reference's DataLoader + `.to(device)` hand to the batch loop (S/train.py:114-115)."""

    def __init__(self, images, masks, patch_size, batch_size, random_shift=False, shuffle=True, seed=0, items=None, shard=None):
        from . import ops
        self.ops = ops
        self.images, self.masks = images, masks
        self.p, self.bs = patch_size, batch_size
        self.shard = shard     # (rank, world): serve this rank's slice [r*B/w, (r+1)*B/w) of every global batch of bs tiles
        if items is not None:  # planned by Loader.plan_items: identical on every rank and on the host path
            self.items = self.chunk_crops = list(items)
            return
        g = torch.Generator().manual_seed(seed)
        items = []
        for i, im in enumerate(images):
            H, W = im.shape[-2:]
            for ty in range(H // patch_size):
                for tx in range(W // patch_size):
                    y, x = ty * patch_size, tx * patch_size
                    if random_shift:   # S/dataset.py:25-26,143: +-20 px; pixels past the border read 0
                        y += int(torch.randint(-20, 21, (1,), generator=g))
                        x += int(torch.randint(-20, 21, (1,), generator=g))
                    items.append((i, y, x))
        if shuffle:
            perm = torch.randperm(len(items), generator=g).tolist()
            items = [items[j] for j in perm]
        self.items = items
        self.chunk_crops = items

    def __len__(self):
        return len(self.items) // self.bs if self.shard else (len(self.items) + self.bs - 1) // self.bs

    def __iter__(self):
        dev = self.images[0].device
        p = self.p
        for b0 in range(0, len(self.items), self.bs):
            batch = self.items[b0:b0 + self.bs]
            if self.shard:
                if len(batch) < self.bs:
                    return      # ragged last global batch: dropped on every rank alike
                r, w = self.shard
                per = self.bs // w
                batch = batch[r * per:(r + 1) * per]
            img = torch.empty((len(batch), 3, p, p), dtype=torch.uint8, device=dev)
            msk = torch.empty((len(batch), p, p), dtype=torch.uint8, device=dev)
            by_image = {}
            for j, (i, y, x) in enumerate(batch):     # one gather launch per source image present in the batch
                by_image.setdefault(i, []).append((j, y, x))
            for i, lst in by_image.items():
                idx = torch.tensor([j for j, _, _ in lst], device=dev)
                org = torch.tensor([[y, x] for _, y, x in lst], dtype=torch.int32, device=dev)
                ti = torch.empty((len(lst), 3, p, p), dtype=torch.uint8, device=dev)
                tm = torch.empty((len(lst), p, p), dtype=torch.uint8, device=dev)
                self.ops.gather_tiles(self.images[i], org, ti)
                self.ops.gather_tiles(self.masks[i], org, tm)
                if len(by_image) == 1:
                    img, msk = ti, tm
                else:
                    img[idx], msk[idx] = ti, tm
            yield img, msk, torch.zeros(1), torch.zeros(1)


def device_class_weights(masks, classes, ignore_background=False):
    """Loader.get_class_weights (S/dataset.py:360-384) with the pixel count done by cvcs_label_histogram on the device"""
    from . import ops
    counts = torch.zeros(classes, dtype=torch.int64, device=masks[0].device)
    for m in masks:
        ops.label_histogram(m.contiguous(), counts)
    return class_weights_from_counts(counts.cpu().float(), ignore_background)


def make_loader(spec, chunk_size, patch_size, num_classes, random_shift=False, seed=0, device=None):
    """`spec` is a dataset directory, or 'synthetic:<n_images>[:<tiles_per_image>]'."""
    if isinstance(spec, str) and spec.startswith("synthetic:"):
        parts = spec.split(":")
        return SyntheticLoader(int(parts[1]), chunk_size, patch_size, num_classes,
                               tiles_per_image=int(parts[2]) if len(parts) > 2 else 16, seed=seed)
    ld = Loader(spec, chunk_size, random_shift=random_shift, patch_size=patch_size, seed=seed)
    ld.device = device
    return ld
