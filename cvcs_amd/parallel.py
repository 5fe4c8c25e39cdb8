"""Data-parallel training: one process per GPU, tiles sharded across ranks, one exchange step per optimiser step.

The reference is single-GPU (`device='cuda:0'`, source/scripts/utils.py:276); this module is new work required by the
north star.  Design for MI355X / xGMI rather than a translation of torch DDP:

  * every parameter gradient already lives in ONE flat f32 buffer laid out in forward order, and the HIP backward
    produces it strictly back-to-front - so a "bucket" is just a contiguous slice [lo, hi) of that buffer and becomes
    ready the moment backward has passed offset lo; no per-parameter hooks, no gradient copies, no re-bucketing;
  * buckets are large (default 32 MiB = 8 Mi floats; the 124 MB of Urnetv2 gradients go out as 4 collectives plus a
    4 MiB one for the front of the buffer, the only slice whose all-reduce cannot overlap backward):
    xGMI is point-to-point (7 links x ~153 GB/s per GPU) and RCCL's ring/tree all-reduce is per-link bound, so few
    large messages beat many small ones;
  * each bucket's all-reduce (RCCL, backend "nccl") is issued on a dedicated HIP stream as soon as it is ready and
    overlaps the rest of backward; the optimiser launch waits on that stream, and the 1/world_size averaging is
    folded into the fused optimiser kernel (grad_scale) instead of a separate pass over the gradients.

By default BatchNorm statistics and the cross-entropy mean stay per-rank (what torch DDP without SyncBN does).
`DataParallel(..., exact=True)` instead reproduces the reference's single-process step on the WHOLE batch (SURVEY
section 8e items 1-2): batch-norm moments and the two backward sums are summed over ranks per layer (tiny f64 / f32
all-reduces in stream order), and the loss is normalised by the number of non-ignored pixels of the whole batch, so
N ranks x B tiles give the same update as one process on N*B tiles up to floating-point summation order.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def plan_buckets(total_floats: int, bucket_floats: int, tail_floats: int | None = None):
    """contiguous [lo, hi) slices covering [0, total), built from the END of the buffer (backward order);
    returned in the order they become ready.

    The slice that starts at offset 0 becomes ready only when backward is over, so its all-reduce is the one nothing
    overlaps: it is kept SMALL (tail_floats, default an eighth of a bucket).  In a U-Net the front of the buffer is
    the cheap-to-send, expensive-to-compute part anyway (levels 1-3 hold 4 % of the parameters and half of the backward
    time), so the big slices before it are long done by then."""
    assert total_floats > 0 and bucket_floats > 0
    if tail_floats is None:
        tail_floats = max(1, bucket_floats // 8)
    tail = min(tail_floats, total_floats) if total_floats > bucket_floats else 0
    out, hi = [], total_floats
    while hi > tail:
        lo = max(tail, hi - bucket_floats)
        out.append((lo, hi))
        hi = lo
    if tail:
        out.append((0, tail))
    return out


class GradientAllReducer:
    def __init__(self, flat_grad: torch.Tensor, bucket_mb: float = 32.0, group=None):
        self.g = flat_grad
        self.group = group
        self.world = dist.get_world_size(group)
        self.buckets = plan_buckets(flat_grad.numel(), max(1, int(bucket_mb * (1 << 20) / 4)))
        self.next = 0
        self.works = []
        self.cuda = flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream(device=flat_grad.device) if self.cuda else None
        self._events = [torch.cuda.Event() for _ in self.buckets] if self.cuda else []      # one per bucket, re-recorded every step

    def begin(self):
        self.next = 0
        self.works = []

    def ready_down_to(self, lo_floats: int, extra_events=()):
        """backward has finished every gradient at flat offset >= lo_floats: launch the buckets that are complete.
        extra_events: events of other streams that produced part of those gradients (the weight-gradient stream)."""
        while self.next < len(self.buckets) and self.buckets[self.next][0] >= lo_floats:
            lo, hi = self.buckets[self.next]
            ev = self._events[self.next] if self.cuda else None
            self.next += 1
            chunk = self.g[lo:hi]
            if self.cuda:
                ev.record(torch.cuda.current_stream())
                self.comm_stream.wait_event(ev)
                for e in extra_events:
                    self.comm_stream.wait_event(e)
                with torch.cuda.stream(self.comm_stream):
                    self.works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:
                self.works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """flush the remaining buckets and make the compute stream wait for all collectives."""
        self.ready_down_to(0)
        for w in self.works:
            w.wait()
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.works = []


class SyncStats:
    """sum-over-ranks of small statistics tensors, ordered with the kernels of the current stream"""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)

    def all_reduce(self, t: torch.Tensor):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)


class DataParallel:
    """Wraps a cvcs_amd network + fused optimiser for N ranks.  Usage (one process per GPU):

        dp = DataParallel(net, opt)       # broadcasts rank 0's parameters / buffers
        loss = crit(net(x_shard), y_shard); opt.zero_grad(); loss.backward(); opt.step()
    """

    def __init__(self, net, optimizer, bucket_mb: float = 32.0, group=None, exact: bool = False, criterion=None):
        """exact=True: SyncBN + whole-batch loss normalisation (pass the CrossEntropyLoss as `criterion`)"""
        self.net, self.opt = net, optimizer
        flat, flat_grad = net.flat_parameters()
        self.world = dist.get_world_size(group)
        dist.broadcast(flat, src=0, group=group)
        for _, b in net.named_buffers():
            if b.dtype == torch.float32:
                dist.broadcast(b, src=0, group=group)
        self.reducer = GradientAllReducer(flat_grad, bucket_mb, group)
        net._engine.on_backward_begin = self.reducer.begin
        net._engine.on_grad_ready = self.reducer.ready_down_to
        optimizer.grad_scale = 1.0 / self.world
        optimizer.pre_step = self.reducer.finish
        if exact:
            assert criterion is not None, "exact=True needs the loss object (its mean spans all ranks' pixels)"
            # the small statistics exchanges get their OWN communicator: on the bucket reducer's they would queue behind
            # whatever 32 MiB all-reduce was issued before them (one internal stream per communicator) and stall the
            # backward pass the buckets are meant to overlap
            sync = SyncStats(dist.new_group(ranks=dist.get_process_group_ranks(group) if group is not None else None))
            net._engine.enable_sync_bn(sync)
            criterion.sync = sync


def shard_batch(global_batch: int, rank: int, world: int):
    """rank r owns tiles [r*B, (r+1)*B) of every global batch."""
    assert global_batch % world == 0
    per = global_batch // world
    return rank * per, (rank + 1) * per
