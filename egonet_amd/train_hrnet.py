"""Native training step of the HRNet heat-map / coordinate model on MI355X
(BASELINE config 4).

Reference hot loop, ``libs/trainer/trainer.py:183-209``::

    optim.zero_grad(); prediction = model(data); loss = loss_func(prediction, target, weights, meta)
    loss.backward(); optim.step()

with ``model`` = PoseHighResolutionNet in train mode (BatchNorm on batch
statistics), ``loss_func`` = JointsCompositeLoss(['mse','l1',None], weights
1.0 / 0.1) (libs/loss/function.py:61-202, KITTI_train_IGRs.yml:86-89) and
Adam(lr 1e-3) (libs/optimizer/optimizer.py:8-40).

``HRNetTrainStep.step`` runs all of that as HIP launches through the C ABI;
torch autograd / MIOpen are not involved:

* forward: the SAME module walk the inference engine records
  (``engine.HRNetEngine._record``) is issued to a tape that executes each layer
  in train mode -- raw conv on the fp32-MFMA kernel (weights packed on the
  device every step), batch statistics + running-stat update, fused
  BatchNorm(+residual)+ReLU -- and remembers what its backward needs;
* backward: the tape is replayed in reverse: fused BatchNorm/ReLU backward
  (two-pass column reductions), weight gradients on the split-K MFMA kernel
  (csrc/conv_wgrad.hip), data gradients on the forward conv kernel with the
  180-degree-rotated, channel-swapped filter (stride 2: over the zero-inserted
  gradient), multi-resolution fuse backward (gated block sums);
* loss: 0.5 * MSE over the maps + 0.1 * L1 over the coordinates, gradients
  written by the loss kernels;
* Adam: ONE launch over the flat parameter buffer (``FlatParams``): parameters,
  gradients and both moments each live in one contiguous allocation, the
  module's ``nn.Parameter``s are views of it (state_dict / HC.pth unchanged).
  The flat gradient is also what a data-parallel job all-reduces (one RCCL
  call per bucket, ``egonet_amd.parallel``).
"""
import contextlib
import gc
import ctypes as C
import os

import torch

from . import _lib, tuner
from .engine import invalidate
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID
from .engine import Buf, HRNetEngine, _round_up


class StepCounters(object):
    """The BatchNorm layers' ``num_batches_tracked`` += 1 and the loss accumulator = 0 as ONE launch at the start of a
    native step (``egn_step_counters_i64``) instead of a ``torch._foreach_add_`` and a ``tensor.zero_()`` [round 6]: the
    device array of the counters' addresses is built once and rebuilt if a buffer moved (``.to()``, ``load_state_dict``
    onto new storage)."""

    def __init__(self):
        self.ptrs = None
        self.table = None

    def tick(self, bns, loss_dev, stream):
        L = _lib.lib()
        ptrs = [bn.num_batches_tracked.data_ptr() for bn in bns if bn.num_batches_tracked is not None]
        if ptrs != self.ptrs:
            dev = loss_dev.device if loss_dev is not None else bns[0].num_batches_tracked.device
            self.table = torch.tensor(ptrs, dtype=torch.int64, device=dev) if ptrs else None
            self.ptrs = ptrs
        _lib.check(L.egn_step_counters_i64(_lib.ptr(self.table), len(ptrs), _lib.ptr(loss_dev), stream), 'step counters')


class FlatParams(object):
    """Trainable parameters as views of one flat fp32 buffer (+ flat grad, m, v)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev = self.params[0].device
        self.offsets = []
        total = 0
        for p in self.params:
            self.offsets.append(total)
            total += _round_up(p.numel(), 4)          # every view stays 16-byte aligned
        self.numel = total
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.flat[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[off:off + p.numel()].view_as(p)
        # step counter and learning rate live in device memory (hipGraph-safe)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self._lr_host = None

    @property
    def t(self):
        return int(self.step_dev.item())

    def adam_step(self, lr, betas, eps, stream, weight_decay=0.0):
        """torch.optim.Adam (optimizer.py:19-21); weight_decay is the coupled L2 form torch implements."""
        if lr != self._lr_host:                 # only when the scheduler changed it (never inside a graph)
            self.lr_dev.fill_(lr)
            self._lr_host = lr
        L = _lib.lib()
        if weight_decay:
            _lib.check(L.egn_adam_l2_step_dev_f32(_lib.ptr(self.flat), _lib.ptr(self.grad), _lib.ptr(self.m),
                                                  _lib.ptr(self.v), self.numel, _lib.ptr(self.lr_dev), betas[0],
                                                  betas[1], eps, weight_decay, _lib.ptr(self.step_dev), stream), 'adam')
        else:
            _lib.check(L.egn_adam_step_dev_f32(_lib.ptr(self.flat), _lib.ptr(self.grad), _lib.ptr(self.m),
                                               _lib.ptr(self.v), self.numel, _lib.ptr(self.lr_dev), betas[0],
                                               betas[1], eps, _lib.ptr(self.step_dev), stream), 'adam')

    def sgd_step(self, lr, momentum, weight_decay, stream):
        """torch.optim.SGD(momentum, weight_decay), dampening 0, no Nesterov (optimizer.py:23-26); the
        momentum buffer lives in ``m``."""
        if lr != self._lr_host:
            self.lr_dev.fill_(lr)
            self._lr_host = lr
        _lib.check(_lib.lib().egn_sgd_step_dev_f32(_lib.ptr(self.flat), _lib.ptr(self.grad), _lib.ptr(self.m),
                                                   self.numel, _lib.ptr(self.lr_dev), momentum, weight_decay,
                                                   _lib.ptr(self.step_dev), stream), 'sgd')

    def update(self, o, stream):
        """One optimizer step as configured on the step object ``o`` (lr, optim_type, betas, eps, momentum,
        weight_decay)."""
        if o.optim_type == 'sgd':
            self.sgd_step(o.lr, o.momentum, o.weight_decay, stream)
        else:
            self.adam_step(o.lr, o.betas, o.eps, stream, o.weight_decay)


@contextlib.contextmanager
def _gc_paused():
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class PackedFilters(object):
    """The packed forward / data-gradient filters of every conv weight of a model.

    The weights change once per iteration (in the optimizer step), so from the second
    step on ALL filters are packed by one launch at the start of the step
    (``egn_pack_conv_weights_batch_f32``) instead of ~600 small ones; the first step
    packs them one by one while it discovers which (weight, direction) pairs exist."""

    _DESC = [('w', '<u8'), ('dst', '<u8'), ('Cout', '<i4'), ('Cin', '<i4'), ('taps', '<i4'), ('dgrad', '<i4'),
             ('begin', '<i8')]

    def __init__(self, device):
        self.dev = device
        self.L = _lib.lib()
        self.entries = {}          # (id(weight), dgrad) -> (weight, packed tensor)
        self.table = None          # device descriptor table once the set is known
        self.total = 0
        self.ptrs = None

    def get(self, weight, dgrad, stream, wino=False):
        """``wino``: the Winograd-transformed filter instead of the direct pack -- True / 1: F(2x2,3x3) (conv_wino.hip,
        egn_conv_config_kind 1), 3: F(4x4,3x3) in conv_wino4.hip's register-feed layout (kind 3)."""
        wino = int(wino)
        code = int(dgrad) | (4 if wino == 3 else (2 if wino else 0))
        ent = self.entries.get((id(weight), code))
        if ent is not None and self.table is not None:
            return ent[1]
        cout, cin, kh, kw = weight.shape
        if ent is None:
            nfl = self.L.egn_wino4_pack_weight_floats(cout, cin, dgrad) if wino == 3 else \
                (self.L.egn_wino_weight_floats(cout, cin, dgrad) if wino else
                 self.L.egn_packed_weight_floats(cout, cin, kh, kw, dgrad))
            if nfl <= 0:
                raise ValueError('no packed layout %d for a %s filter' % (wino, tuple(weight.shape)))
            wp = torch.empty(nfl, dtype=torch.float32, device=self.dev)
            self.entries[(id(weight), code)] = (weight, wp)
            self.table = None
        else:
            wp = ent[1]
        if wino == 3:
            _lib.check(self.L.egn_wino4_pack_weight_f32(_lib.ptr(weight), cout, cin, dgrad, _lib.ptr(wp), stream),
                       'wino4 pack')
        elif wino:
            _lib.check(self.L.egn_wino_pack_weight_f32(_lib.ptr(weight), cout, cin, dgrad, _lib.ptr(wp), stream),
                       'wino pack')
        else:
            _lib.check(self.L.egn_pack_conv_weight_f32(_lib.ptr(weight), cout, cin, kh, kw, dgrad, _lib.ptr(wp),
                                                       stream), 'pack')
        return wp

    def _pointers(self):
        return [w.data_ptr() for (w, _) in self.entries.values()]

    def finalize(self):
        """Build the device descriptor table for the (weight, direction) pairs seen so far."""
        import numpy as np
        if self.table is not None or not self.entries:
            return
        desc = np.zeros(len(self.entries), dtype=np.dtype(self._DESC, align=True))
        assert desc.dtype.itemsize == self.L.egn_pack_desc_bytes(), (desc.dtype.itemsize, self.L.egn_pack_desc_bytes())
        begin = 0
        for i, ((_, code), (w, wp)) in enumerate(self.entries.items()):
            cout, cin, kh, kw = w.shape
            desc[i] = (w.data_ptr(), wp.data_ptr(), cout, cin, kh * kw, code, begin)
            begin += wp.numel() // (48 if code & 4 else (64 if code & 2 else 4))     # work units (egonet_hip.h)
        self.total = begin
        self.table = torch.from_numpy(desc.view(np.uint8)).to(self.dev)
        self.ptrs = self._pointers()

    def pack_all(self, stream):
        """One launch for every filter; False if the table is not built yet (first step) or a
        parameter was re-allocated since (``.to()`` / ``load_state_dict`` on a new storage)."""
        if self.table is None:
            return False
        if self._pointers() != self.ptrs:
            self.table = None
            return False
        _lib.check(self.L.egn_pack_conv_weights_batch_f32(_lib.ptr(self.table), len(self.entries), self.total,
                                                          stream), 'pack all')
        return True


class _Tape(object):
    """Recorder interface of ``engine._Recorder`` that EXECUTES train-mode layers."""

    def __init__(self, owner, images):
        self.o = owner
        self.L = owner.L
        self.dev = images.device
        self.images = images
        self.st = _lib.current_stream(self.dev)
        self.data = {}            # id(Buf) -> 1-D fp32 tensor
        self.grad = {}            # id(Buf) -> [tensor, owned]
        self.keep = []            # Bufs (ids stay unique while the tape lives)
        self.side = owner.wgrad_stream      # weight gradients run here, beside the backward chain
        self.side_st = None if self.side is None else C.c_void_p(self.side.cuda_stream)
        self.side_keep = []       # tensors the side stream reads: alive until it is joined
        self.back = []            # backward closures, forward order
        self.named = {}           # tag -> Buf
        self.user = {}            # tag -> NCHW copy for the caller
        self.no_grad = set()      # ids of Bufs that need no gradient (the input)
        self.bns = []
        self.maps_user = None

    # -- recorder plumbing (concurrency hints are ignored: one stream) ------
    def fork(self):
        pass

    def join(self):
        pass

    def lane(self, k):
        pass

    def decode(self, *a, **k):
        raise NotImplementedError('decode is not part of the training step')

    def _empty(self, numel):
        return torch.empty(numel, dtype=torch.float32, device=self.dev)

    def new(self, n, h, w, c, cs=None, name=''):
        b = Buf(n, h, w, c, cs=cs, name=name)
        self.keep.append(b)
        self.data[id(b)] = self._empty(n * h * w * b.cs)
        return b

    def _accum(self, buf, g, owned=True):
        if id(buf) in self.no_grad:
            return
        cur = self.grad.get(id(buf))
        if cur is None:
            self.grad[id(buf)] = [g, owned]
            return
        if not cur[1]:                              # shared tensor: do not add in place
            out = self._empty(g.numel())
            _lib.check(self.L.egn_add_f32(_lib.ptr(cur[0]), _lib.ptr(g), _lib.ptr(out), g.numel(), self.st), 'add')
            self.grad[id(buf)] = [out, True]
        else:
            _lib.check(self.L.egn_add_f32(_lib.ptr(cur[0]), _lib.ptr(g), _lib.ptr(cur[0]), g.numel(), self.st), 'add')

    def _take_grad(self, buf):
        cur = self.grad.pop(id(buf), None)
        return None if cur is None else cur[0]

    # -- launches -----------------------------------------------------------
    def _pack(self, weight, dgrad):
        return self.o.packs.get(weight, dgrad, self.st)

    def _conv_launch(self, x, wp, shift, y, n, h, w, cin, cs_in, cout, cs_out, kh, kw, stride, pad, act,
                     weight=None, dgrad=0, want_stats=False, res=None, f43_ok=True):
        """``wp``: the direct-packed filter (None: packed here from ``weight``).  With ``weight`` (+ ``dgrad``) given, the tuner may pick a
        Winograd configuration for 3x3 stride-1 layers (forward and data gradient alike: the data
        gradient is a stride-1 convolution with the rotated filter); the transformed filter then comes
        from the step's PackedFilters like the direct one."""
        key = (n, h, w, cin, cs_in, cout, cs_out, kh, kw, stride, pad, False, False)
        can_wino = weight is not None and act in (ACT_NONE, ACT_RELU) and self.o.allow_wino
        # [round 5] F(4x4,3x3) (csrc/conv_wino4.hip, filter kind 3) in the tape: the filter is transformed on the device
        # with all the others (PackedFilters), BatchNorm statistics come from conv_wino4s_kernel's item end, the K-split
        # configurations get the owner's ticket words (one stream: launches that share them are ordered)
        can_f43 = can_wino and f43_ok and self.o.allow_f43 in (('all',) if dgrad else ('all', 'fwd'))
        cfg = tuner.choose(self.dev, key, allow_wino=can_wino, allow_f43=can_f43)
        kind = self.L.egn_conv_config_kind(cfg) if cfg > 0 else 0
        if kind == 2:
            # conv_wino43_kernel's filter layout (kind 2) is not one the tape packs: never launch it on a direct-packed
            # filter -- take the best configuration of the kinds the tape feeds instead
            cfg = tuner.choose(self.dev, key, allow_wino=can_wino, allow_f43=False)
            kind = self.L.egn_conv_config_kind(cfg) if cfg > 0 else 0
        ntk = 0
        if kind == 3:
            ntk = self.L.egn_conv2d_ticket_words(n, h, w, cin, cs_in, cout, cs_out, kh, kw, stride, pad, cfg)
            if ntk > 0 and (ntk > self.o.tickets.numel() or (res is not None and res.data_ptr() == y.data_ptr())):
                # (the K split writes a raw share into y before the residual is read: not for the in-place gradient add)
                cfg = tuner.choose(self.dev, key, allow_wino=can_wino, allow_f43=False)
                kind = self.L.egn_conv_config_kind(cfg) if cfg > 0 else 0
                ntk = 0
        if kind == 3:
            wp = self.o.packs.get(weight, dgrad, self.st, wino=3)
        elif can_wino and kind == 1:
            wp = self.o.packs.get(weight, dgrad, self.st, wino=True)
        elif wp is None:
            wp = self._pack(weight, dgrad)
        stats = None
        if want_stats and self.o.fuse_bn_stats and cfg > 0:
            # BatchNorm batch statistics in the conv epilogue (per-tile partial sums, csrc/conv_wino.hip):
            # saves the separate reduction pass over z where the tile configuration supports it
            nrows = self.L.egn_conv2d_bnstats_rows(n, h, w, cin, cs_in, cout, cs_out, kh, kw, stride, pad, cfg)
            if nrows > 0:
                stats = (torch.empty(nrows * 2 * cout, dtype=torch.float64, device=self.dev), nrows)
        tm = self.o.timing
        if tm is not None:         # bench.py: hipEvents around every forward / data-gradient conv launch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.dev))
        if kind == 3:
            _lib.check(self.L.egn_conv2d_ex_f32(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(self.o.ones), _lib.ptr(shift),
                                                _lib.ptr(res), _lib.ptr(y), n, h, w, cin, cs_in, cout, cs_out, kh, kw,
                                                stride, pad, act, cfg, None if stats is None else _lib.ptr(stats[0]),
                                                0 if stats is None else stats[1],
                                                _lib.ptr(self.o.tickets) if ntk > 0 else None, self.o.tickets.numel(),
                                                self.st), 'conv (F(4x4,3x3))')
        elif stats is not None:
            _lib.check(self.L.egn_conv2d_bnstats_f32(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(self.o.ones), _lib.ptr(shift),
                                                     _lib.ptr(y), n, h, w, cin, cs_in, cout, cs_out, kh, kw, stride,
                                                     pad, cfg, _lib.ptr(stats[0]), stats[1], self.st), 'conv+stats')
        else:
            _lib.check(self.L.egn_conv2d_f32(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(self.o.ones), _lib.ptr(shift),
                                             _lib.ptr(res), _lib.ptr(y), n, h, w, cin, cs_in, cout, cs_out, kh, kw,
                                             stride, pad, act, 0, cfg, self.st), 'conv')
        if tm is not None:
            e1.record(torch.cuda.current_stream(self.dev))
            ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
            tm.append((cfg, 2.0 * n * ho * wo * cout * cin * kh * kw, e0, e1, stats is not None and kind == 3))
        return stats

    def _wgrad(self, x, xd, dy, cs_out, weight, stride, pad):
        cout, cin, kh, kw = weight.shape
        L = self.L
        need = L.egn_conv2d_wgrad_ws_bytes(x.n, x.h, x.w, cin, x.cs, cout, cs_out, kh, kw, stride, pad)
        if need < 0:
            raise NotImplementedError('weight gradient of a %dx%d convolution' % (kh, kw))
        ws = self.o.wgrad_ws(need)
        st = self.st
        if self.side is not None:
            # nothing on the backward chain waits for a weight gradient (only the optimizer does):
            # issue it on the side stream once dy exists, so the latency-bound BatchNorm / reduction
            # kernels of the chain overlap its MFMA work.  One side stream: the launches share ws.
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.dev))
            self.side.wait_event(ev)
            self.side_keep.append((xd, dy))
            st = self.side_st
        _lib.check(L.egn_conv2d_wgrad_f32(_lib.ptr(xd), _lib.ptr(dy), _lib.ptr(self.o.grad_of(weight)), x.n, x.h, x.w, cin, x.cs,
                                          cout, cs_out, kh, kw, stride, pad, _lib.ptr(ws), ws.numel() * 4, st),
                   'wgrad')

    def release(self):
        """Break the tape <-> backward-closure reference cycles so that every tensor of the iteration is
        freed by reference counting when the step returns (left to the cyclic garbage collector they
        linger for a few iterations, the caching allocator has to hipMalloc fresh blocks meanwhile and
        single steps take 2x as long)."""
        self.back = []
        self.data = {}
        self.grad = {}
        self.keep = []
        self.named = {}
        self.side_keep = []

    def join_side(self):
        """The side stream's weight gradients are complete for everything issued after this on the
        main stream (gradient all-reduce, Adam); the tensors they read may be released."""
        if self.side is not None and self.side_keep:
            ev = torch.cuda.Event()
            ev.record(self.side)
            torch.cuda.current_stream(self.dev).wait_event(ev)
            self.side_keep = []

    def _accum_dgrad(self, x, dy, ho, wo, cs_out, weight, stride, pad):
        """grad(x) += data gradient of the conv.  Where x already has a gradient tensor of its own (the
        residual path of a block reaches x first) the conv adds it in its epilogue and writes in place
        (every element is read and written by the same lane) instead of a separate add pass."""
        if id(x) in self.no_grad:
            return
        cur = self.grad.get(id(x))
        into = cur[0] if (cur is not None and cur[1] and weight.shape[2] == weight.shape[3] and self.o.fuse_grad_add) else None
        g = self._dgrad(dy, ho, wo, cs_out, weight, stride, pad, x, into=into)
        if into is None:
            self._accum(x, g)

    def _dgrad(self, dy, ho, wo, cs_out, weight, stride, pad, x, into=None):
        cout, cin, kh, kw = weight.shape
        L = self.L
        if kh != kw:
            # the Pedestrian model's final 4x3 'valid' conv over the whole 4x3 map (hrnet.py:457-460):
            # one output pixel, so dx[n, (ky,kx), ci] = sum_co dy[n, co] * W[co, ci, ky, kx] is a GEMM
            # whose output rows are the NHWC map itself
            if not (pad == 0 and stride == 1 and x.h == kh and x.w == kw and ho == 1 and wo == 1):
                raise NotImplementedError('data gradient of a non-square %dx%d convolution that does not '
                                          'cover its whole input' % (kh, kw))
            wt = torch.zeros(kh * kw, x.cs, cout, dtype=torch.float32, device=self.dev)
            wt[:, :cin] = weight.detach().permute(2, 3, 1, 0).reshape(kh * kw, cin, cout)
            wt = wt.view(kh * kw * x.cs, cout, 1, 1)
            rows = kh * kw * x.cs
            wq = torch.empty(L.egn_packed_weight_floats(rows, cout, 1, 1, 0), dtype=torch.float32, device=self.dev)
            _lib.check(L.egn_pack_conv_weight_f32(_lib.ptr(wt), rows, cout, 1, 1, 0, _lib.ptr(wq), self.st), 'pack')
            dx = self._empty(x.n * rows)
            shift = self.o.zeros if self.o.zeros.numel() >= rows + 16 else torch.zeros(rows + 16, device=self.dev)
            ones = self.o.ones if self.o.ones.numel() >= rows + 16 else torch.ones(rows + 16, device=self.dev)
            cfg = tuner.choose(self.dev, (x.n, 1, 1, cout, cs_out, rows, rows, 1, 1, 1, 0, False, False))
            _lib.check(L.egn_conv2d_f32(_lib.ptr(dy), _lib.ptr(wq), _lib.ptr(ones), _lib.ptr(shift), None, _lib.ptr(dx),
                                        x.n, 1, 1, cout, cs_out, rows, rows, 1, 1, 1, 0, ACT_NONE, 0, cfg, self.st),
                       'conv')
            return dx
        if stride == 2:
            up = self._empty(x.n * x.h * x.w * cs_out)
            _lib.check(L.egn_zero_insert2_f32(_lib.ptr(dy), _lib.ptr(up), x.n, ho, wo, x.h, x.w, cs_out, self.st),
                       'zero_insert')
            src, sh, sw = up, x.h, x.w
        elif stride == 1:
            src, sh, sw = dy, ho, wo
        else:
            raise NotImplementedError('stride %d' % stride)
        # a stride-2 conv's data gradient is a stride-1 conv over the zero-inserted dy: Winograd applies too
        dx = into if into is not None else self._empty(x.n * x.h * x.w * x.cs)
        # (F(4x4,3x3) only for the stride-1 layers: over a zero-inserted gradient its per-launch error measured
        # 2.0-2.5e-5 of the largest element at 32 crops -- above the 2e-5 the float64 launch checks allow -- against
        # <= 1.4e-5 on the stride-1 layers; tests/test_gpu_bench_size.py, profiles/r5_train32_dgrad_errors.txt)
        self._conv_launch(src, None, self.o.zeros, dx, x.n, sh, sw, cout, cs_out, cin, x.cs, kh, kw, 1, kh - 1 - pad,
                          ACT_NONE, weight=weight, dgrad=1, res=into, f43_ok=(stride == 1))
        return dx

    # -- ops (engine._Recorder interface) -----------------------------------
    def nchw_to_nhwc(self, x_ext, n, c, h, w, tag=''):
        y = self.new(n, h, w, c, name=tag)
        _lib.check(self.L.egn_nchw_to_nhwc_f32(_lib.ptr(self.images), _lib.ptr(self.data[id(y)]), n, c, h, w, y.cs,
                                               self.st), 'to_nhwc')
        self.no_grad.add(id(y))
        return y

    def nhwc_to_nchw(self, x, c, dst_ext, tag=''):
        self.maps_user = torch.empty(x.n, c, x.h, x.w, dtype=torch.float32, device=self.dev)
        _lib.check(self.L.egn_nhwc_to_nchw_f32(_lib.ptr(self.data[id(x)]), _lib.ptr(self.maps_user), x.n, c, x.h, x.w,
                                               x.cs, self.st), 'to_nchw')

    def ramps(self, y, c0, tag=''):
        _lib.check(self.L.egn_fill_coord_ramps_f32(_lib.ptr(self.data[id(y)]), y.n, y.h, y.w, y.cs, c0, self.st),
                   'ramps')

    def conv(self, x, weight, bias=None, bn=None, act=ACT_NONE, res=None, stride=1, pad=0,
             dst=None, out_nchw=False, cout_cs=None, tag=''):
        L = self.L
        cout, cin, kh, kw = weight.shape
        assert cin == x.c, (cin, x.c, tag)
        ho = (x.h + 2 * pad - kh) // stride + 1
        wo = (x.w + 2 * pad - kw) // stride + 1
        # user-facing outputs (dst / out_nchw of the inference recording) are computed
        # into an internal NHWC tensor like every other layer -- the backward needs the
        # padded layout -- and copied out in the caller's NCHW format afterwards
        z = self.new(x.n, ho, wo, cout, cs=cout_cs, name=tag)
        xd, zd = self.data[id(x)], self.data[id(z)]
        wp = None                        # packed by _conv_launch in the layout its tile configuration reads
        rows = x.n * ho * wo
        if bn is None:
            shift = self.o.zeros
            if bias is not None:
                shift = torch.zeros(_round_up(cout, 16), dtype=torch.float32, device=self.dev)
                shift[:cout].copy_(bias.detach())
            self._conv_launch(xd, wp, shift, zd, x.n, x.h, x.w, cin, x.cs, cout, z.cs, kh, kw, stride, pad, act,
                              weight=weight)
            if dst is not None or out_nchw:
                u = torch.empty(x.n, cout, ho, wo, dtype=torch.float32, device=self.dev)
                _lib.check(L.egn_nhwc_to_nchw_f32(_lib.ptr(zd), _lib.ptr(u), x.n, cout, ho, wo, z.cs, self.st), 'to_nchw')
                self.user[tag] = u
            self.named[tag] = z

            def backward():
                dy = self._take_grad(z)
                if dy is None:
                    return
                if act == ACT_SIGMOID:
                    dz = self._empty(dy.numel())
                    _lib.check(L.egn_sigmoid_bwd_f32(_lib.ptr(dy), _lib.ptr(zd), _lib.ptr(dz), dy.numel(), self.st),
                               'sigmoid_bwd')
                    dy = dz
                elif act != ACT_NONE:
                    raise NotImplementedError('activation %d without BatchNorm' % act)
                if bias is not None and bias.requires_grad:
                    _lib.check(L.egn_colsum_f32(_lib.ptr(dy), rows, cout, z.cs, _lib.ptr(self.o.grad_of(bias)),
                                                _lib.ptr(self.o.col_ws), self.st), 'bias grad')
                if weight.requires_grad:
                    self._wgrad(x, xd, dy, z.cs, weight, stride, pad)
                self._accum_dgrad(x, dy, ho, wo, z.cs, weight, stride, pad)
            backward.params = [weight] + ([bias] if bias is not None else [])     # gradients final after it
            self.back.append(backward)
            return z

        if bias is not None:
            raise NotImplementedError('conv bias followed by BatchNorm')
        stats = self._conv_launch(xd, wp, self.o.zeros, zd, x.n, x.h, x.w, cin, x.cs, cout, z.cs, kh, kw, stride, pad,
                                  ACT_NONE, weight=weight, want_stats=True)
        mean, istd = self._empty(cout), self._empty(cout)
        mom = 0.1 if bn.momentum is None else bn.momentum
        if stats is not None:      # the conv epilogue wrote per-tile partial sums: only the finalise stage is left
            _lib.check(L.egn_bn_stats_finalize_f32(_lib.ptr(stats[0]), stats[1], rows, cout, bn.eps, _lib.ptr(mean),
                                                   _lib.ptr(istd), None, _lib.ptr(bn.running_mean),
                                                   _lib.ptr(bn.running_var), mom, self.st), 'bn_stats_finalize')
        else:
            _lib.check(L.egn_bn_stats_f32(_lib.ptr(zd), rows, cout, z.cs, bn.eps, _lib.ptr(mean), _lib.ptr(istd), None,
                                          _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var), mom,
                                          _lib.ptr(self.o.col_ws), self.st), 'bn_stats')
        self.bns.append(bn)
        y = self.new(x.n, ho, wo, cout, cs=z.cs, name=tag)
        yd = self.data[id(y)]
        relu = 1 if act == ACT_RELU else 0
        if act not in (ACT_NONE, ACT_RELU):
            raise NotImplementedError('activation %d after BatchNorm' % act)
        rd = None if res is None else self.data[id(res)]
        _lib.check(L.egn_bn_act_fwd_f32(_lib.ptr(zd), _lib.ptr(mean), _lib.ptr(istd), _lib.ptr(bn.weight),
                                        _lib.ptr(bn.bias), None, 1.0, relu, _lib.ptr(rd), _lib.ptr(yd), rows, cout,
                                        z.cs, self.st), 'bn_act_fwd')
        self.named[tag] = y

        def backward():
            dy = self._take_grad(y)
            if dy is None:
                return
            dbeta = self.o.grad_of(bn.bias) if bn.bias.requires_grad else self._empty(cout)
            dgamma = self.o.grad_of(bn.weight) if bn.weight.requires_grad else self._empty(cout)
            _lib.check(L.egn_bn_bwd_sums_f32(_lib.ptr(dy), _lib.ptr(zd), None, 1.0, _lib.ptr(mean), _lib.ptr(istd),
                                             _lib.ptr(bn.weight), _lib.ptr(bn.bias), relu, _lib.ptr(rd), rows, cout,
                                             z.cs, _lib.ptr(dbeta), _lib.ptr(dgamma), _lib.ptr(self.o.col_ws),
                                             self.st), 'bn_bwd_sums')
            dz = self._empty(dy.numel())
            dres = self._empty(dy.numel()) if (res is not None and id(res) not in self.no_grad) else None
            _lib.check(L.egn_bn_bwd_dz_f32(_lib.ptr(dy), _lib.ptr(zd), None, 1.0, _lib.ptr(mean), _lib.ptr(istd),
                                           _lib.ptr(bn.weight), _lib.ptr(bn.bias), relu, _lib.ptr(rd), _lib.ptr(dbeta),
                                           _lib.ptr(dgamma), _lib.ptr(dz), _lib.ptr(dres), rows, cout, z.cs, self.st),
                       'bn_bwd_dz')
            if self.o.debug_hook is not None:
                self.o.debug_hook(dict(tag=tag, dy=dy, z=zd, mean=mean, istd=istd, bn=bn, res=rd, dbeta=dbeta,
                                       dgamma=dgamma, dz=dz, dres=dres, relu=relu, rows=rows, cols=cout, ld=z.cs))
            if dres is not None:
                self._accum(res, dres)
            if weight.requires_grad:
                self._wgrad(x, xd, dz, z.cs, weight, stride, pad)
            self._accum_dgrad(x, dz, ho, wo, z.cs, weight, stride, pad)
        backward.params = [weight, bn.weight, bn.bias]
        self.back.append(backward)
        return y

    def fuse(self, terms, relu, tag=''):
        L = self.L
        base = [t for t, s in terms if s == 0][0]
        y = self.new(base.n, base.h, base.w, base.c, cs=base.cs, name=tag)
        yd = self.data[id(y)]
        nt = len(terms)
        ptrs = (C.c_void_p * nt)(*[self.data[id(t)].data_ptr() for t, _ in terms])
        shifts = (C.c_int * nt)(*[s for _, s in terms])
        _lib.check(L.egn_fuse_sum_relu_f32(_lib.ptr(yd), y.n, y.h, y.w, y.c, y.cs, nt, ptrs, shifts, int(relu), self.st),
                   'fuse')
        self.named[tag] = y

        def backward():
            dy = self._take_grad(y)
            if dy is None:
                return
            gate = _lib.ptr(yd) if relu else None
            shared = {}
            for t, s in terms:
                if id(t) in self.no_grad:
                    continue
                g = shared.get(s)
                if g is None:
                    g = self._empty(t.n * t.h * t.w * t.cs)
                    _lib.check(L.egn_fuse_bwd_f32(_lib.ptr(dy), gate, _lib.ptr(g), y.n, y.h, y.w, y.cs, s, self.st),
                               'fuse_bwd')
                    shared[s] = g
                self._accum(t, g, owned=False)       # same-shift terms share one tensor
        self.back.append(backward)
        return y


class TapeOwner(object):
    """What a ``_Tape`` needs from the object that drives it: the library, scratch vectors, the packed-filter
    cache, the weight-gradient side stream and workspace, the kernel-family switches, and ``grad_of(param)`` =
    the tensor a parameter's gradient kernels write.  Two owners: ``HRNetTrainStep`` (the whole iteration
    natively, gradients in one flat buffer) and ``egonet_amd.autograd.HRNetAutograd`` (forward / backward of a
    ``torch.autograd.Function``: torch owns loss and optimiser)."""

    def _init_tape_owner(self, model):
        p0 = next(model.parameters())
        if not p0.is_cuda:
            raise ValueError('%s needs the model on a GPU' % type(self).__name__)
        if model.head_type not in ('coordinates', 'heatmap') or model.pixel_shuffle:
            raise NotImplementedError('native training: head_type %r pixel_shuffle %r'
                                      % (model.head_type, model.pixel_shuffle))
        self.model = model
        self.dev = p0.device
        self.L = _lib.lib()
        widest = max(p.shape[0] for p in model.parameters()) + 32
        self.ones = torch.ones(_round_up(widest, 16), dtype=torch.float32, device=self.dev)
        self.zeros = torch.zeros(_round_up(widest, 16), dtype=torch.float32, device=self.dev)
        self.col_ws = torch.zeros(self.L.egn_colreduce_ws_bytes(widest) // 4, dtype=torch.float32, device=self.dev)
        self._wgrad_ws = None
        # weight gradients on a second stream (EGONET_AMD_WGRAD_STREAM=0: everything on one stream)
        # EGONET_AMD_WGRAD_PRIORITY: HIP stream priority of that stream (0 normal, positive = lower where the runtime
        # has a low level): the chain's latency-bound BatchNorm / reduction kernels should not queue behind the
        # weight gradients' grids
        self.wgrad_stream = torch.cuda.Stream(device=self.dev, priority=int(os.environ.get('EGONET_AMD_WGRAD_PRIORITY', '0'))) \
            if os.environ.get('EGONET_AMD_WGRAD_STREAM', '1') != '0' else None
        self.walker = HRNetEngine(model)
        self.packs = PackedFilters(self.dev)
        self.debug_hook = None        # tests/train_debug.py: per-layer checks of the BatchNorm backward
        self.timing = None            # bench.py: a list collects (cfg, flops, start, end) per conv launch
        # 3x3 stride-1 forward / data-gradient convolutions may run on the fused Winograd kernels
        # (csrc/conv_wino.hip) where they measured faster (EGONET_AMD_TRAIN_WINO=0: direct kernels only)
        self.allow_wino = os.environ.get('EGONET_AMD_TRAIN_WINO', '1') != '0'
        # ... and on the F(4x4,3x3) kernels (csrc/conv_wino4.hip) [round 5]: EGONET_AMD_TRAIN_F43 = all (forward and
        # data-gradient convolutions), fwd (forward only), 0 (F(2x2,3x3) / direct only)
        f43 = os.environ.get('EGONET_AMD_TRAIN_F43', 'all')
        self.allow_f43 = {'1': 'all', 'all': 'all', 'fwd': 'fwd'}.get(f43, '0')
        # ticket words of the K-split configurations (cfg 83 / 84): zero between launches, shared by the launches of
        # the tape's one stream
        self.tickets = torch.zeros(1 << 16, dtype=torch.int32, device=self.dev)
        torch.cuda.current_stream(self.dev).synchronize()      # (zero in memory before any stream's first launch)
        self.fuse_bn_stats = os.environ.get('EGONET_AMD_FUSE_BN_STATS', '1') != '0'
        self.fuse_grad_add = os.environ.get('EGONET_AMD_FUSE_GRAD_ADD', '1') != '0'

    def grad_of(self, p):
        return p.grad

    def wgrad_ws(self, nbytes):
        if self._wgrad_ws is None or self._wgrad_ws.numel() * 4 < nbytes:
            # allocated in the pool of the stream that uses it: when it has to grow, the old block is
            # only handed to later work of that same stream
            with torch.cuda.stream(self.wgrad_stream if self.wgrad_stream is not None
                                   else torch.cuda.current_stream(self.dev)):
                self._wgrad_ws = torch.empty(nbytes // 4 + 1024, dtype=torch.float32, device=self.dev)
        return self._wgrad_ws

class HRNetTrainStep(TapeOwner):
    """``step(images, target, joints_xy)`` = one iteration of trainer.py:183-209."""

    CR_CRITERIA = {'mse': 0, 'l1': 1, 'sl1': 2}      # loss_dict, function.py:17-20

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, w_hm=1.0, w_coor=0.1, grad_sync=None,
                 sigma=1, w_cr=None, cr_type='sl1', cr_indices=None, target_cr=4.0 / 3.0, cr_loss_thres=0.15,
                 hm_type='mse', coor_type='l1', optim_type='adam', momentum=0.0, weight_decay=0.0,
                 use_target_weight=False):
        self._init_tape_owner(model)
        p0 = next(model.parameters())
        if model.head_type == 'heatmap' and w_coor:
            raise NotImplementedError("the 'heatmap' head trains with the heat-map term only (w_coor=0)")
        # JointsMSELoss(use_target_weight) (function.py:22-46, the heat-map head's criterion): both maps are multiplied
        # by target_weight[:, k] before the per-joint MSE.  JointsCompositeLoss (the coordinate head) stores the flag and
        # never uses it (function.py:95-111), so there it changes nothing -- like the reference.
        self.use_target_weight = bool(use_target_weight) and model.head_type == 'heatmap'
        if self.use_target_weight and hm_type != 'mse':
            raise NotImplementedError('JointsMSELoss is an MSE criterion (function.py:24-26)')
        self.lr, self.betas, self.eps = lr, betas, eps
        if optim_type not in ('adam', 'sgd'):
            raise NotImplementedError('optimizer %r (optimizer.py:8-40 knows adam and sgd)' % (optim_type,))
        self.optim_type, self.momentum, self.weight_decay = optim_type, float(momentum), float(weight_decay)
        # criteria of the heat-map and coordinate terms: any of loss_dict (function.py:17-20)
        self.hm_crit, self.coor_crit = self.CR_CRITERIA[hm_type], self.CR_CRITERIA[coor_type]
        self.w_hm, self.w_coor = float(w_hm), float(w_coor or 0.0)
        # cross-ratio term (function.py:113-153, train_IGRs.py:44-46): off unless a weight is
        # given ('None' in the shipped YAML) AND apply_cr_loss is set (trainer.py:168-169:
        # from the second epoch on)
        self.w_cr = None if w_cr in (None, 'None') else float(w_cr)
        self.apply_cr_loss = False
        self.cr_idx = None
        if self.w_cr is not None:
            if model.head_type != 'coordinates':
                raise NotImplementedError('the cross-ratio term needs the coordinate head')
            if cr_indices is None:
                from .common.img_proc import CR_INDICES_BBOX12
                cr_indices = CR_INDICES_BBOX12
            idx = torch.as_tensor(cr_indices, dtype=torch.int32).reshape(-1, 4)
            if int(idx.min()) < 0 or int(idx.max()) >= model.num_joints:
                raise ValueError('cr_indices outside 0..%d' % (model.num_joints - 1))
            self.cr_idx = idx.contiguous().to(p0.device)
            self.cr_crit = self.CR_CRITERIA[cr_type]
            self.target_cr, self.cr_loss_thres = float(target_cr), float(cr_loss_thres)
        self.grad_sync = grad_sync
        self.sigma = sigma           # heatmapModel.sigma: targets drawn on the device when step(target=None)
        self.last_target_weight = None
        self.flat = FlatParams(model.parameters())
        self.loss_dev = torch.zeros(1, dtype=torch.float64, device=self.dev)
        self.counters = StepCounters()
        self.last_maps = self.last_coords = None

    @torch.no_grad()
    def step(self, images, target, joints_xy=None, update=True, joints_vis=None, target_weight=None):
        """images [N,3,H,W], target [N,K,h,w] heat-maps (None: drawn on the device from
        joints_xy / joints_vis with ``self.sigma``), joints_xy [N,K,2] in input pixels
        (``meta['transformed_joints'][:, :, :2]``).  Returns the loss as a 1-element
        float64 device tensor (no host sync).

        The cyclic garbage collector is paused while the ~1 500 launches of the iteration are issued: a
        full collection in the middle (35 ms measured, every ~10 iterations) starves the GPU and doubles
        that step's time; between steps it hides behind the queued work.  Nothing here needs it -- the
        tape's reference cycles are broken explicitly (``_Tape.release``)."""
        with _gc_paused():
            return self._step(images, target, joints_xy, update, joints_vis, target_weight)

    def _step(self, images, target, joints_xy, update, joints_vis, target_weight=None):
        m, L = self.model, self.L
        if not m.training:
            raise RuntimeError('HRNetTrainStep.step needs model.train()')
        images = images.contiguous().float()
        if target is not None:
            target = target.contiguous().float()
        n, cin, h, w = images.shape
        if h % 32 or w % 32:
            raise ValueError('HRNet input height/width must be multiples of 32, got %dx%d' % (h, w))
        with torch.cuda.device(self.dev):
            st = _lib.current_stream(self.dev)
            self.flat.grad.zero_()
            self.packs.pack_all(st)          # every forward / data-gradient filter, one launch
            tape = _Tape(self, images)
            self.walker._record(n, cin, h, w, None, r=tape)
            J = m.num_joints
            self.counters.tick(tape.bns, self.loss_dev, st)       # num_batches_tracked += 1, loss = 0: one launch
            if m.head_type == 'coordinates':
                aug, coords = tape.named['head1'], tape.named['head2.4']
                cd = tape.user['head2.4'].view(n, 2 * J)          # compact [N, 2K] = coords [N,K,2]
                self.last_coords = cd.view(n, J, 2)
                self.last_maps = tape.maps_user
                use_cr = self.w_cr is not None and self.apply_cr_loss
                if self.w_coor or use_cr:
                    dc = tape._empty(cd.numel())
                    if self.w_coor:
                        if joints_xy is None:
                            raise ValueError('the coordinate term needs joints_xy')
                        gt = torch.as_tensor(joints_xy, dtype=torch.float32).to(self.dev)[..., :2].clone()
                        gt[..., 0] /= w            # function.py:160-161 (img_size = (width, height))
                        gt[..., 1] /= h
                        gt = gt.contiguous()
                        _lib.check(L.egn_elem_loss_f32(_lib.ptr(cd), _lib.ptr(gt), 1, cd.numel(), cd.numel(),
                                                       cd.numel(), self.coor_crit, self.w_coor, 0, _lib.ptr(dc),
                                                       _lib.ptr(self.loss_dev), st), 'coor loss')
                    else:
                        dc.zero_()
                    if use_cr:
                        nl = self.cr_idx.shape[0]
                        ws = tape._empty(L.egn_cross_ratio_ws_bytes(n, nl) // 4)
                        _lib.check(L.egn_cross_ratio_f32(_lib.ptr(cd), n, J, _lib.ptr(self.cr_idx), nl,
                                                         self.target_cr, self.cr_loss_thres, self.cr_crit, self.w_cr,
                                                         _lib.ptr(dc), _lib.ptr(self.loss_dev), _lib.ptr(ws), st),
                                   'cross_ratio')
                    dpad = tape._empty(n * coords.cs)             # back to the padded NHWC row layout
                    _lib.check(L.egn_nchw_to_nhwc_f32(_lib.ptr(dc), _lib.ptr(dpad), n, 2 * J, 1, 1, coords.cs, st))
                    tape.grad[id(coords)] = [dpad, True]
            else:
                aug = tape.named['final_layer']
                self.last_maps = tape.user['final_layer']
            if target is None:
                # heat-map targets drawn on the device from the joints (img_proc.py:347-409):
                # the [N,K,h,w] target never crosses PCIe
                if joints_xy is None:
                    raise ValueError('step() needs target heat-maps or joints_xy to draw them from')
                if aug.h != aug.w or h != w:
                    raise NotImplementedError('device-side targets: square maps only (the reference mixes the '
                                              'width/height indices of input_size / heatmap_size, img_proc.py:376-383)')
                from .common import img_proc
                target, self.last_target_weight = img_proc.generate_target_batch(
                    joints_xy, torch.ones(n, J) if joints_vis is None else joints_vis,
                    dict(target_type='gaussian', input_size=(w, h), heatmap_size=(aug.h, aug.w), sigma=self.sigma),
                    device=self.dev)
            if tuple(target.shape) != (n, J, aug.h, aug.w):
                raise ValueError('target must be %s, got %s' % ((n, J, aug.h, aug.w), tuple(target.shape)))
            tg = tape._empty(n * aug.h * aug.w * aug.cs)
            _lib.check(L.egn_nchw_to_nhwc_f32(_lib.ptr(target), _lib.ptr(tg), n, J, aug.h, aug.w, aug.cs, st))
            da = torch.zeros(n * aug.h * aug.w * aug.cs, dtype=torch.float32, device=self.dev)
            pred_flat = tape.data[id(aug)]
            wv = None
            if self.use_target_weight:
                # 0.5 * mean((pred * w - gt * w)^2): the same kernel on the weighted maps, gradient * w afterwards
                # (three broadcast multiplies over [N,h,w,K]; an option no shipped configuration switches on)
                tw = target_weight if target_weight is not None else self.last_target_weight
                if tw is None:
                    raise ValueError('use_target_weight needs target_weight [N,K(,1)] (or device-drawn targets)')
                wv = torch.zeros(n, 1, 1, aug.cs, dtype=torch.float32, device=self.dev)
                wv[:, 0, 0, :J] = torch.as_tensor(tw, dtype=torch.float32).reshape(n, J).to(self.dev)
                pred_flat = (pred_flat.view(n, aug.h, aug.w, aug.cs) * wv).reshape(-1)
                tg = (tg.view(n, aug.h, aug.w, aug.cs) * wv).reshape(-1)
            # (1/K) sum_k 0.5*crit_k = 0.5 * crit over all joints (equal element counts), function.py:95-111
            _lib.check(L.egn_elem_loss_f32(_lib.ptr(pred_flat), _lib.ptr(tg), n * aug.h * aug.w, J, aug.cs,
                                           aug.cs, self.hm_crit, 0.5 * self.w_hm, 0, _lib.ptr(da),
                                           _lib.ptr(self.loss_dev), st), 'hm loss')
            if wv is not None:
                da = (da.view(n, aug.h, aug.w, aug.cs) * wv).reshape(-1)
            tape._accum(aug, da)
            # the gradient all-reduce of a slice of the flat buffer starts (on a communication
            # stream) as soon as every parameter in it has its gradient kernels issued
            sess = None
            if hasattr(self.grad_sync, 'begin'):
                # a parameter written by several closures (shared weights) is final after its LAST report
                counts = {}
                for fn in tape.back:
                    for q in getattr(fn, 'params', ()):
                        counts[id(q)] = counts.get(id(q), 0) + 1
                sess = self.grad_sync.begin(self.flat, torch.cuda.current_stream(self.dev), self.wgrad_stream,
                                            report_counts=counts)
            for fn in reversed(tape.back):
                fn()
                if sess is not None:
                    sess.done(getattr(fn, 'params', ()))
            tape.join_side()
            if sess is not None:
                sess.finish()
            elif self.grad_sync is not None:
                self.grad_sync(self.flat.grad)
            if update:
                self.flat.update(self, st)
            self.packs.finalize()     # first step: the set of filters is known now
            invalidate(m)             # the inference engine caches folded weights (raw-pointer writes)
            if self.debug_hook is not None:
                self.last_tape = tape
            else:
                tape.release()
        return self.loss_dev
