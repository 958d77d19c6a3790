"""One process per GPU: sharding of crops / key-point sets across ranks and the
single exchange step of data-parallel training.

The reference's only parallelism is ``torch.nn.DataParallel`` inside one
process (tools/train_IGRs.py:59,111): per step it re-broadcasts all parameters,
scatters the batch, gathers outputs to GPU 0 and reduces gradients there.  Here
every rank owns a full weight copy and a contiguous shard of the batch:

  inference  independent replicas, NO collective on the data path; only the
             per-instance results (a few hundred bytes per crop) are gathered
             when the caller asks for them (``gather_results``);
  training   one exchange per step: all-reduce(mean) of the gradients, packed
             into a few large flat buckets (xGMI links are point-to-point, so
             few large messages beat many small ones), launched bucket by
             bucket as soon as the backward pass has produced them.

Backend ``nccl`` is RCCL on ROCm; ``gloo`` is used by the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """Contiguous [lo, hi) of ``total`` items for ``rank``; the first
    ``total % world`` ranks get one extra item (ragged shards are allowed)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_records(records, world, rank):
    """Shard per-instance records by rank keeping instances of one image
    together when possible is NOT required by the path (instances are
    independent); plain contiguous split."""
    lo, hi = shard_range(len(records), world, rank)
    return records[lo:hi]


def gather_results(local, group=None):
    """All-gather a dict of per-instance tensors with ragged first dimension.
    Returns the concatenation in rank order on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    out = {}
    for k in sorted(local):
        t = local[k].contiguous()
        n = torch.tensor([t.shape[0]], dtype=torch.long, device=t.device)
        counts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(counts, n, group=group)
        counts = [int(c.item()) for c in counts]
        m = max(counts)
        pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        out[k] = torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
    return out


class GradBuckets(object):
    """Flat-bucket gradient all-reduce (mean) for data-parallel training.

    Parameters are packed in REVERSE registration order (the order in which
    backward produces their gradients) into buckets of ``bucket_mb``; each
    bucket is reduced with one collective.  ``reduce()`` is called after
    ``loss.backward()``; with ``async_op`` the collectives of early buckets
    overlap the packing of later ones.
    """

    def __init__(self, params, bucket_mb=32.0, group=None):
        self.group = group
        self.params = [p for p in params if p.requires_grad]
        cap = int(bucket_mb * 2 ** 20)
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > cap:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)

    def reduce(self):
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(self.group)
        if world == 1:
            return
        pending = []
        for bucket in self.buckets:
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
            flat = torch.cat([g.reshape(-1) for g in grads])
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            pending.append((work, flat, bucket))
        for work, flat, bucket in pending:
            work.wait()
            flat.div_(world)
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    p.grad = flat[off:off + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[off:off + n].view_as(p))
                off += n


class FlatGradSync(object):
    """Gradient all-reduce (mean) over ONE flat gradient buffer, in place, in
    slices of ``bucket_mb`` -- the ``grad_sync`` hook of the native training steps
    (egonet_amd.train_hrnet.FlatParams keeps every gradient in one allocation,
    so nothing is packed or copied).

    The slices are issued back to front: the native backward fills the flat
    buffer from its END (last layers first), so with ``async_op`` the early
    collectives overlap what is still being reduced.  Replaces the per-step
    parameter broadcast + output gather of ``torch.nn.DataParallel``
    (tools/train_IGRs.py:59) by the one exchange step data parallelism needs.
    xGMI is point to point: 32 MB slices keep every ring step well above the
    latency floor without serialising the whole 256 MB HRNet gradient.
    """

    def __init__(self, bucket_mb=32.0, group=None):
        self.group = group
        self.bucket = max(1, int(bucket_mb * 2 ** 20) // 4)

    def slices(self, numel):
        out, hi = [], numel
        while hi > 0:
            lo = max(0, hi - self.bucket)
            out.append((lo, hi))
            hi = lo
        return out

    def __call__(self, flat):
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(self.group)
        if world == 1:
            return
        pending = []
        for lo, hi in self.slices(flat.numel()):
            pending.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for work in pending:
            work.wait()
        flat.div_(world)


def broadcast_module(module, src=0, group=None):
    """Make every rank start from rank ``src``'s parameters and buffers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
