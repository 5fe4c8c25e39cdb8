"""CPU coverage of the N > 1 path: bucket planning and the gradient reducer over gloo, world_size 2."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cvcs_amd.parallel import GradientAllReducer, plan_buckets, shard_batch


def test_plan_buckets_cover_back_to_front():
    b = plan_buckets(1000, 300, tail_floats=0)
    assert b == [(700, 1000), (400, 700), (100, 400), (0, 100)]
    assert plan_buckets(10, 100) == [(0, 10)]
    # the slice that is ready last (it starts at offset 0) is the small one: nothing overlaps its all-reduce
    assert plan_buckets(1000, 300, tail_floats=50) == [(700, 1000), (400, 700), (100, 400), (50, 100), (0, 50)]
    total = 31044496 + 64  # Urnetv2 parameters (+ alignment padding)
    bk = plan_buckets(total, 8 << 20)
    assert bk[0][1] == total and bk[-1][0] == 0 and all(a[0] == b_[1] for a, b_ in zip(bk, bk[1:]))
    assert len(bk) == 5 and bk[-1] == (0, 1 << 20) and all(hi - lo <= 8 << 20 for lo, hi in bk)


def test_shard_batch():
    assert [shard_batch(64, r, 4) for r in range(4)] == [(0, 16), (16, 32), (32, 48), (48, 64)]
    with pytest.raises(AssertionError):
        shard_batch(10, 0, 4)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1000
        g = torch.arange(n, dtype=torch.float32) * (rank + 1)          # rank r holds (r+1) * arange
        red = GradientAllReducer(g, bucket_mb=300 * 4 / (1 << 20))     # 300-float buckets -> 4 collectives
        launched = []
        red.begin()
        # backward reports progress back-to-front at "layer" boundaries that do not coincide with bucket edges
        for lo in (950, 640, 400, 399, 120):
            red.ready_down_to(lo)
            launched.append(red.next)
        red.finish()
        expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        ok = torch.equal(g, expect)
        # second step reuses the reducer
        g.copy_(torch.ones(n) * (rank + 1))
        red.begin()
        red.finish()
        ok2 = torch.equal(g, torch.full((n,), float(sum(r + 1 for r in range(world)))))
        q.put((rank, launched, ok, ok2))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gradient_allreducer_gloo(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, launched, ok, ok2 in res:
        # buckets (700,1000) (400,700) (100,400) (0,100): ready after offsets 950->0, 640->1, 400->2, 399->2, 120->2
        assert launched == [0, 1, 2, 2, 2], (rank, launched)
        assert ok and ok2
