"""Tile producer host logic (CPU): the reference's `Loader` API (source/scripts/dataset.py:228-387) with every random decision
drawn from (seed, epoch, chunk), so that the ranks of a data-parallel job partition each epoch exactly, and
the reference's zero-filling crop for shifted tiles (v2.functional.crop, S/dataset.py:25-32,143)."""
import os

import numpy as np
import pytest
import torch

from cvcs_amd import dataset, utils


@pytest.fixture()
def gid_dir(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    for sub in ("Image__8bit_NirRGB", "Annotation__index", "Annotation__color"):
        os.makedirs(tmp_path / sub)
    for i in range(3):
        Image.fromarray(rng.integers(0, 256, (448, 672, 3), dtype=np.uint8)).save(tmp_path / "Image__8bit_NirRGB" / f"im{i}.png")
        Image.fromarray(rng.integers(0, 6, (448, 672), dtype=np.uint8)).save(tmp_path / "Annotation__index" / f"im{i}.png")
    return str(tmp_path)


def test_crop_zero_filled_matches_padded_crop():
    t = torch.arange(2 * 10 * 12, dtype=torch.uint8).view(2, 10, 12)
    pad = torch.zeros(2, 10 + 40, 12 + 40, dtype=torch.uint8)
    pad[:, 20:30, 20:32] = t
    for y, x in ((-5, -7), (0, 0), (6, 9), (-20, 3), (9, 11), (3, -1)):
        assert torch.equal(dataset.crop_zero_filled(t, y, x, 4), pad[:, 20 + y:24 + y, 20 + x:24 + x])
    m = t[0]
    assert torch.equal(dataset.crop_zero_filled(m, 8, 10, 4), pad[0, 28:32, 30:34])


def test_loader_is_deterministic_in_seed_epoch_chunk_and_ranks_partition_the_epoch(gid_dir):
    """two processes = two Loader objects with the same seed: identical image order, tile order and shifts in every epoch;
    the per-rank shards of chunk_batches are disjoint and their union is the single-process batch"""
    mk = lambda: dataset.Loader(gid_dir, chunk_size=2, random_shift=True, patch_size=224, seed=7)   # noqa: E731
    a, b, single = mk(), mk(), mk()
    assert a.tpi == 6
    for epoch in range(2):
        for ld in (a, b, single):
            ld.shuffle()
        assert a.idxs == b.idxs
        for c in range(len(a)):
            assert a.plan_items(c) == b.plan_items(c)
            whole = list(utils.chunk_batches(single, c, 4, "cpu"))
            r0 = list(utils.chunk_batches(a, c, 4, "cpu", shard=(0, 2)))
            r1 = list(utils.chunk_batches(b, c, 4, "cpu", shard=(1, 2)))
            full = [w for w in whole if w[0].shape[0] == 4]          # the ragged last global batch is dropped on every rank
            assert len(r0) == len(r1) == len(full)
            for w, x0, x1 in zip(full, r0, r1):
                assert torch.equal(torch.cat([x0[0], x1[0]]), w[0]) and torch.equal(torch.cat([x0[1], x1[1]]), w[1])
    # different epochs / seeds give different orders
    assert dataset.Loader(gid_dir, 2, True, 224, seed=8).plan_items(0) != mk().plan_items(0)


def test_shifted_tiles_zero_fill_like_the_reference_crop(gid_dir):
    ld = dataset.Loader(gid_dir, chunk_size=3, random_shift=True, patch_size=224, seed=1)
    items = ld.plan_items(0)
    chunk = ld.get_iterable_chunk(0)
    decoded = [ld._decode(i) for i in ld.chunks[0]]
    outside = 0
    for (pos, y, x), (img, mask, _, _) in zip(items, chunk):
        im, mk = decoded[pos]
        for yy in (0, 223):
            for xx in (0, 223):
                sy, sx = y + yy, x + xx
                inside = 0 <= sy < ld.H and 0 <= sx < ld.W
                outside += not inside
                assert torch.equal(img[:, yy, xx], im[:, sy, sx] if inside else torch.zeros(3, dtype=torch.uint8))
                assert mask[yy, xx] == (mk[sy, sx] if inside else 0)     # label 0 = the ignored class
    assert outside > 0   # the fixture does exercise the border


def test_class_weights_from_the_loader(gid_dir):
    ld = dataset.Loader(gid_dir, 2, patch_size=224)
    w = ld.get_class_weights(6, ignore_background=True)
    counts = torch.zeros(6)
    for i in range(3):
        counts += torch.bincount(ld._decode(i)[1].reshape(-1).long(), minlength=6).float()
    assert torch.allclose(w, dataset.class_weights_from_counts(counts, True)) and w[0] == 0


def test_random_tps_adds_rescaled_crops_of_the_full_images(gid_dir):
    """`random_tps = [(aug_size, percentage)]` (S/dataset.py:173-198, read at S/train.py:109): int(percentage * tiles) extra crops of
    aug_size x aug_size pixels from random images of the chunk, image resized bilinearly (antialiased) and index mask by nearest-exact to the
    patch size, shuffled in - planned from (seed, epoch, chunk): two loaders serve the same tiles; the synthetic loader (no full images) refuses"""
    mk = lambda: dataset.Loader(gid_dir, chunk_size=2, random_shift=False, patch_size=224, seed=5)   # noqa: E731
    a, b = mk(), mk()
    base = list(a.get_iterable_chunk(0))
    aug = list(a.get_iterable_chunk(0, random_tps=[(300, 0.5), (100, 0.25)]))
    aug_b = list(b.get_iterable_chunk(0, random_tps=[(300, 0.5), (100, 0.25)]))
    n = len(base)
    assert len(aug) == n + int(0.5 * n) + int(0.25 * n)
    assert all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(aug, aug_b))
    assert all(t[0].shape == (3, 224, 224) and t[0].dtype == torch.uint8 and t[1].shape == (224, 224) and t[1].dtype == torch.uint8 for t in aug)
    # the plain tiles are all still there; the extra ones are resized crops: every one equals resize_tile of SOME planned window
    key = lambda t: (t[0].sum().item(), t[1].sum().item())   # noqa: E731
    plain = {key(t) for t in base}
    extra = [t for t in aug if key(t) not in plain]
    assert len(extra) == int(0.5 * n) + int(0.25 * n)
    import random
    rng = random.Random(((5 * 1000003 + 0) * 1000003 + 0) * 31 + 7)
    plan = dataset.plan_random_tps([(300, 0.5), (100, 0.25)], n, 2, a.H, a.W, rng)
    decoded = [a._decode(i) for i in a.chunks[0]]
    want = {key(dataset.resize_tile(decoded[pos][0][:, y:y + s, x:x + s], decoded[pos][1][y:y + s, x:x + s], 224)) for pos, y, x, s in plan}
    assert {key(t) for t in extra} == want
    # a mask resized by nearest-exact only contains labels of its window
    for pos, y, x, s in plan[:3]:
        _, m = dataset.resize_tile(decoded[pos][0][:, y:y + s, x:x + s], decoded[pos][1][y:y + s, x:x + s], 224)
        assert set(m.unique().tolist()) <= set(decoded[pos][1][y:y + s, x:x + s].unique().tolist())
    with pytest.raises(ValueError):
        a.get_iterable_chunk(0, random_tps=[(1000, 0.1)])          # larger than the image
    syn = dataset.SyntheticLoader(4, 2, 64, 5, tiles_per_image=4, seed=1)
    with pytest.raises(ValueError):
        syn.get_iterable_chunk(0, random_tps=[(32, 0.5)])
