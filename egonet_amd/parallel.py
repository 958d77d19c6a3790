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


class _SliceLayout(object):
    """Which slices of the flat gradient each parameter overlaps, and how many parameters each slice waits
    for -- a property of (FlatGradSync.bucket, FlatParams layout): computed ONCE and copied per step (it
    used to be rebuilt every iteration: O(parameters x slices) of Python inside the step's launch loop)."""

    def __init__(self, sync, flat_params):
        numel = flat_params.grad.numel()
        self.key = (sync.bucket, numel, len(flat_params.params))
        self.slices = sync.slices(numel)
        self.spans = {}
        self.waiting = [0] * len(self.slices)
        bounds = [lo for lo, _ in self.slices]         # descending starts: slice i covers [bounds[i], bounds[i-1])
        for p, off in zip(flat_params.params, flat_params.offsets):
            end = off + p.numel()
            hit = [i for i, (lo, hi) in enumerate(self.slices) if off < hi and end > lo]
            self.spans[id(p)] = tuple(hit)
            for i in hit:
                self.waiting[i] += 1
        del bounds


class _SyncSession(object):
    """One backward pass of ``FlatGradSync.begin``: tracks which slices of the flat gradient are
    final and launches their all-reduce while the rest of the backward still runs.

    A parameter's gradient is final when every backward closure that writes it has reported it.  By
    default that is ONE report per parameter (HRNet / the lifter: every weight belongs to one layer);
    ``report_counts`` ({id(param): n}) declares parameters written by several closures (shared weights).  A
    report for a parameter whose slice is already in flight -- the gradient would be reduced while it is
    still being written -- raises instead of corrupting the step."""

    def __init__(self, sync, flat_params, main_stream, side_stream, report_counts=None):
        self.sync, self.flat = sync, flat_params
        self.grad = flat_params.grad
        self.world = dist.get_world_size(sync.group)
        self.main, self.side = main_stream, side_stream
        self.cuda = self.grad.is_cuda
        lay = sync.layout(flat_params)
        self.slices = lay.slices
        self.spans = lay.spans                         # read-only; per-step state is `left` / `waiting`
        self.waiting = list(lay.waiting)
        self.left = {}                                 # id(param) -> reports still expected
        if report_counts:
            for pid, n in report_counts.items():
                if n > 1 and pid in self.spans:
                    self.left[pid] = n
        self.reported = set()
        self.launched = [False] * len(self.slices)
        self.works = []
        if self.cuda and sync.comm_stream is None:
            sync.comm_stream = torch.cuda.Stream(device=self.grad.device, priority=sync.comm_priority)

    def _launch(self, i):
        lo, hi = self.slices[i]
        self.launched[i] = True
        if not self.cuda:
            self.works.append(dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.sync.group,
                                              async_op=True))
            return
        comm = self.sync.comm_stream
        # the slice is final once everything issued so far on the backward's streams has run
        for k, st in enumerate((self.main, self.side)):
            if st is not None:
                ev = self.sync.event(i, k)
                ev.record(st)
                comm.wait_event(ev)
        with torch.cuda.stream(comm):
            self.works.append(dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.sync.group,
                                              async_op=True))

    def done(self, params):
        """The kernels that write the gradients of ``params`` have been issued."""
        for p in params:
            pid = id(p)
            span = self.spans.get(pid)
            if span is None:                       # not a trainable parameter of the flat buffer (frozen)
                continue
            if pid in self.reported:
                raise RuntimeError('FlatGradSync: a backward closure reported a parameter (%s) whose gradient was '
                                   'already declared final -- its slice of the flat gradient may be in flight.  '
                                   'Declare parameters written by several closures with report_counts, or set '
                                   'EGONET_AMD_GRAD_OVERLAP=0.' % (tuple(p.shape),))
            n = self.left.get(pid, 1) - 1
            if n > 0:
                self.left[pid] = n
                continue
            self.reported.add(pid)
            for i in span:
                self.waiting[i] -= 1
                if self.waiting[i] == 0 and not self.launched[i]:
                    self._launch(i)

    def finish(self):
        """Launch what is left (parameters the backward never reported), wait, take the mean."""
        for i in range(len(self.slices)):
            if not self.launched[i]:
                self._launch(i)
        for w in self.works:
            w.wait()                    # CUDA: the current (main) stream waits for the collective
        self.grad.div_(self.world)


class FlatGradSync(object):
    """Gradient all-reduce (mean) over ONE flat gradient buffer, in place, in
    slices of ``bucket_mb`` -- the ``grad_sync`` hook of the native training steps
    (egonet_amd.train_hrnet.FlatParams keeps every gradient in one allocation,
    so nothing is packed or copied).

    Two ways to use it:
      * ``sync(flat_grad)`` after the backward: every slice, back to front, then the mean;
      * ``sess = sync.begin(flat_params, main_stream, side_stream)`` before the backward,
        ``sess.done([params...])`` whenever the kernels writing those gradients have been issued,
        ``sess.finish()`` at the end: a slice is all-reduced ON A COMMUNICATION STREAM as soon as
        every parameter in it is final -- the native backward fills the buffer from its end (last
        layers first), so the collectives of the late layers run under the MFMA work of the early
        ones and only the first layers' slice is exposed.  The order in which slices become final
        is a property of the model, identical on every rank, so the collectives match up.
        ``EGONET_AMD_GRAD_OVERLAP=0`` makes ``begin`` return None: the steps then fall back to the
        first form (one exchange after the backward) -- the switch to pull if the overlapped path
        misbehaves on a new fabric.
    Replaces the per-step parameter broadcast + output gather of ``torch.nn.DataParallel``
    (tools/train_IGRs.py:59) by the one exchange step data parallelism needs.
    xGMI is point to point: 32 MB slices keep every ring step well above the
    latency floor without serialising the whole 256 MB HRNet gradient.

    The communication stream is created with HIGH priority (``comm_priority=-1``;
    ``EGONET_AMD_COMM_PRIORITY`` overrides): an RCCL ring kernel occupies a few CUs per channel, the
    backward's MFMA kernels fill all 256 -- at equal priority a collective that becomes ready in the
    middle of the backward only starts when a compute kernel drains; with priority its workgroups are
    dispatched as soon as CUs free up and the exchange really runs under the backward.  RCCL settings this
    path is written for (none are set by the package): ``NCCL_MIN_NCHANNELS`` / ``NCCL_MAX_NCHANNELS`` left
    at their defaults (ring over the 7 xGMI links of a fully connected 8-GPU node), ``HSA_ENABLE_IPC_MODE_LEGACY=0``
    (dmabuf IPC, required by this driver), ``RCCL_MSCCL_ENABLE`` untouched; ``tools/scale_check.sh`` prints the
    exposed (non-overlapped) all-reduce time so that a regression of the overlap is visible.
    """

    def __init__(self, bucket_mb=32.0, group=None, comm_priority=None):
        import os
        self.group = group
        self.bucket = max(1, int(bucket_mb * 2 ** 20) // 4)
        self.comm_stream = None
        if comm_priority is None:
            comm_priority = int(os.environ.get('EGONET_AMD_COMM_PRIORITY', '-1'))
        self.comm_priority = comm_priority
        self.overlap = os.environ.get('EGONET_AMD_GRAD_OVERLAP', '1') != '0'
        self._layout = None
        self._events = {}

    def active(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def layout(self, flat_params):
        key = (self.bucket, flat_params.grad.numel(), len(flat_params.params))
        if self._layout is None or self._layout.key != key or self._layout_owner is not flat_params:
            self._layout = _SliceLayout(self, flat_params)
            self._layout_owner = flat_params
        return self._layout

    def event(self, i, k):
        """Re-used event (slice i, stream k): a slice launches once per step, after last step's wait."""
        ev = self._events.get((i, k))
        if ev is None:
            ev = self._events[(i, k)] = torch.cuda.Event()
        return ev

    def begin(self, flat_params, main_stream=None, side_stream=None, report_counts=None):
        """-> session (see the class docstring), or None when there is nothing to reduce / the overlap is
        switched off (the caller then runs ``sync(flat.grad)`` after the backward)."""
        if not (self.active() and self.overlap):
            return None
        return _SyncSession(self, flat_params, main_stream, side_stream, report_counts)

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


def broadcast_buffers(module, src=0, group=None):
    """BatchNorm running statistics follow rank ``src``: what ``torch.nn.DataParallel`` does implicitly
    (only the replica on device 0 shares its buffers with the module the caller keeps,
    tools/train_IGRs.py:59).  During training every rank updates its own running statistics from its
    shard (they are not read in training mode); call this before evaluating or saving on other ranks."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in module.buffers():
        dist.broadcast(t.data, src=src, group=group)


def broadcast_module(module, src=0, group=None):
    """Make every rank start from rank ``src``'s parameters and buffers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
