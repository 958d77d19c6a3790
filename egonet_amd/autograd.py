"""torch.autograd bridge: the reference's UNCHANGED training loop on the native kernels.

The reference's hot loop (``libs/trainer/trainer.py:183-209``)::

    optim.zero_grad(); prediction = model(data); loss = loss_func(prediction, target, weights, meta)
    loss.backward(); optim.step()

calls the module in train mode on CUDA tensors under autograd.  ``HRNetAutograd`` / ``LifterAutograd`` make
that call ONE ``torch.autograd.Function`` node whose forward runs the native train-mode tape (fp32-MFMA /
Winograd convolutions, BatchNorm on batch statistics incl. the running-statistics update -- the kernels of
``egonet_amd.train_hrnet`` / ``train_lifter``) and whose backward replays the tape in reverse (BatchNorm/ReLU
backward, weight gradients on the side stream, data gradients, fuse backward) from the output gradients
torch hands in.  torch keeps what the reference's loop owns: the loss (``JointsCompositeLoss`` /
``MSELoss1D`` on the returned tensors), ``.grad`` accumulation and the optimiser.  MIOpen / rocBLAS never
run.  The fully native steps (``HRNetTrainStep`` / ``LifterTrainStep``: loss + Adam as HIP kernels, flat
gradient buffer, overlapped all-reduce) stay the faster path; this is the drop-in one.

Limits (raise or fall back loudly, never silently): the input must not require a gradient (the modules
route such calls to the torch graph); heads as in ``HRNetTrainStep`` ('coordinates', 'heatmap').
"""
import torch

from . import _lib
from .engine import invalidate
from .train_hrnet import TapeOwner, _Tape, _gc_paused
from .train_lifter import LifterTrainStep


class _HRNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bridge, x, *params):
        tape, outs = bridge._forward(x)
        ctx.bridge, ctx.tape = bridge, tape
        ctx.nparam = len(params)
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gouts):
        tape, ctx.tape = ctx.tape, None
        if tape is None:
            raise RuntimeError('egonet_amd HRNet autograd node: backward called twice (the tape is released after '
                               'the first backward, like retain_graph=False)')
        grads = ctx.bridge._backward(tape, gouts)
        return (None, None) + tuple(grads)


class HRNetAutograd(TapeOwner):
    """``bridge(x)`` = ``model(x)`` in train mode under autograd, on the native kernels."""

    def __init__(self, model):
        self._init_tape_owner(model)
        self._grads = None

    def params(self):
        return [p for p in self.model.parameters() if p.requires_grad]

    def grad_of(self, p):
        return self._grads[id(p)]

    def __call__(self, x):
        if x.requires_grad:
            raise NotImplementedError('the native tape does not produce the gradient of the input crops')
        return _HRNetFn.apply(self, x, *self.params())

    @torch.no_grad()
    def _forward(self, images):
        m = self.model
        images = images.contiguous().float()
        n, cin, h, w = images.shape
        if h % 32 or w % 32:
            raise ValueError('HRNet input height/width must be multiples of 32, got %dx%d' % (h, w))
        with _gc_paused(), torch.cuda.device(self.dev):
            st = _lib.current_stream(self.dev)
            self.packs.pack_all(st)          # every forward / data-gradient filter, one launch (2nd call on)
            tape = _Tape(self, images)
            self.walker._record(n, cin, h, w, None, r=tape)
            torch._foreach_add_([bn.num_batches_tracked for bn in tape.bns], 1)
            invalidate(m)                    # BatchNorm running statistics were written through raw pointers
            J = m.num_joints
            if m.head_type == 'coordinates':
                outs = (tape.maps_user, tape.user['head2.4'].view(n, J, 2))
            else:
                outs = (tape.user['final_layer'],)
        return tape, outs

    @torch.no_grad()
    def _backward(self, tape, gouts):
        m, L = self.model, self.L
        params = self.params()
        with _gc_paused(), torch.cuda.device(self.dev):
            st = _lib.current_stream(self.dev)
            # one fresh flat buffer per backward; the views go back to autograd, which accumulates them into
            # (or, when .grad is None, adopts them as) the parameters' .grad
            sizes = [(p.numel() + 3) // 4 * 4 for p in params]
            flat = torch.zeros(sum(sizes), dtype=torch.float32, device=self.dev)
            self._grads, views, off = {}, [], 0
            for p, sz in zip(params, sizes):
                v = flat[off:off + p.numel()].view_as(p)
                self._grads[id(p)] = v
                views.append(v)
                off += sz
            J = m.num_joints
            n = tape.images.shape[0]
            if m.head_type == 'coordinates':
                aug, coords = tape.named['head1'], tape.named['head2.4']
                g_maps, g_coords = gouts
                if g_coords is not None:
                    dc = g_coords.contiguous().float().view(n, 2 * J)
                    dpad = tape._empty(n * coords.cs)             # back to the padded NHWC row layout
                    _lib.check(L.egn_nchw_to_nhwc_f32(_lib.ptr(dc), _lib.ptr(dpad), n, 2 * J, 1, 1, coords.cs, st))
                    tape.grad[id(coords)] = [dpad, True]
            else:
                aug = tape.named['final_layer']
                g_maps = gouts[0]
            if g_maps is not None:
                gm = g_maps.contiguous().float()
                da = tape._empty(n * aug.h * aug.w * aug.cs)
                _lib.check(L.egn_nchw_to_nhwc_f32(_lib.ptr(gm), _lib.ptr(da), n, J, aug.h, aug.w, aug.cs, st))
                tape._accum(aug, da)
            for fn in reversed(tape.back):
                fn()
            tape.join_side()
            self.packs.finalize()     # first backward: the set of (filter, direction) pairs is known now
            if self.debug_hook is not None:
                self.last_tape = tape
            else:
                tape.release()
            self._grads = None
        return views


class _Pending(object):
    """One train-mode forward whose backward has not run.  The bridge's in-kernel dropout needs "at most one forward
    pending"; a forward whose graph is DROPPED without a backward (a metrics forward in train mode, an exception
    between forward and backward, a discarded micro-batch) must release its claim too, or the bridge would fall back
    to torch-generated masks for the rest of its life: the token lives in the autograd node's context and
    gives the count back when that context dies, whichever way."""
    __slots__ = ('bridge', 'live')

    def __init__(self, bridge):
        self.bridge, self.live = bridge, True
        bridge._pending += 1

    def release(self):
        if self.live:
            self.live = False
            self.bridge._pending = max(0, self.bridge._pending - 1)

    def __del__(self):
        self.release()


class _LifterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bridge, x, *params):
        pred, saved = bridge._fwd(x)
        ctx.bridge, ctx.saved = bridge, saved
        ctx.token = bridge._last_token
        bridge._last_token = None
        return pred

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        saved, ctx.saved = ctx.saved, None
        if saved is None:
            raise RuntimeError('egonet_amd lifter autograd node: backward called twice')
        try:
            return (None, None) + tuple(ctx.bridge._bwd(saved, gout))
        finally:
            ctx.token.release()


class LifterAutograd(LifterTrainStep):
    """``bridge(x)`` = ``FCModel(x)`` in train mode under autograd, on the native kernels (the GEMMs,
    BatchNorm1d / ReLU / dropout forward and backward of ``LifterTrainStep``; torch owns loss and optimiser)."""

    def __init__(self, model):
        # the parent's machinery without its flat parameter buffer / optimiser state: parameters stay
        # ordinary tensors that torch's optimiser updates
        import os
        from .train_hrnet import PackedFilters
        from .train_lifter import _Unit
        from .engine import _round_up
        p0 = next(model.parameters())
        if not p0.is_cuda:
            raise ValueError('LifterAutograd needs the model on a GPU')
        self.model = model
        self.dev = p0.device
        self.act = 2 if model.leaky else 1
        self.grad_sync = None
        self.units = [_Unit(model.w1, model.batch_norm1)]
        for blk in model.res_blocks:
            self.units += [_Unit(blk.w1, blk.batch_norm1), _Unit(blk.w2, blk.batch_norm2)]
        self.final = model.w2
        self.packs = PackedFilters(p0.device)
        self.w4 = {}
        for fc in [u.fc for u in self.units] + [self.final]:
            self.w4[id(fc.weight)] = fc.weight.detach().view(fc.out_features, fc.in_features, 1, 1)
        self._ws = {}
        self._wgrad_floats = 0
        widest = _round_up(max([u.outf for u in self.units] + [u.inf for u in self.units]
                               + [self.final.out_features]), 16) + 16
        self.ones = torch.ones(widest, dtype=torch.float32, device=self.dev)
        self.zeros = torch.zeros(widest, dtype=torch.float32, device=self.dev)
        self.L = _lib.lib()
        self.wgrad_stream = torch.cuda.Stream(device=self.dev) \
            if os.environ.get('EGONET_AMD_WGRAD_STREAM', '1') != '0' else None
        self._side_used = False
        self._side_keep = []
        self.timing = None
        self._grads = None
        # in-kernel dropout: forward and backward of one call share (seed, unit, counter value); the counter is a
        # device tensor bumped once per forward -- a backward runs before the next forward of the reference's
        # loop; with several forwards in flight (test: two forwards, two backwards) the masks are torch tensors
        self.drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self.drop_step = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._pending = 0
        self._last_token = None

    @property
    def p(self):                      # the module's dropout probability, live (nn.Dropout.p may be edited)
        return float(self.model.p_dropout)

    def params(self):
        return [p for p in self.model.parameters() if p.requires_grad]

    def grad_of(self, p):
        return self._grads[id(p)]

    def __call__(self, x):
        if x.requires_grad:
            raise NotImplementedError('the native lifter tape does not produce the gradient of its input')
        if any(not p.requires_grad for p in self.model.parameters()):
            raise NotImplementedError('frozen lifter parameters under the autograd bridge')
        return _LifterFn.apply(self, x, *self.params())

    @torch.no_grad()
    def _fwd(self, x):
        # filter views follow the parameters (an optimiser may have re-bound .data)
        for fc in [u.fc for u in self.units] + [self.final]:
            w4 = self.w4.get(id(fc.weight))
            if w4 is None or w4.data_ptr() != fc.weight.data_ptr():
                self.w4[id(fc.weight)] = fc.weight.detach().view(fc.out_features, fc.in_features, 1, 1)
        with _gc_paused(), torch.cuda.device(self.dev):
            # the counter value must still be this forward's when its backward runs: only one forward may be pending
            self.rng_dropout = LifterTrainStep.rng_dropout and self._pending == 0
            if self.rng_dropout:
                self.drop_step.add_(1)
            self._last_token = _Pending(self)          # handed to the autograd node's context by _LifterFn.forward
            pred, saved = self._forward(x, fresh=True)
            invalidate(self.model)
        return pred, saved

    @torch.no_grad()
    def _bwd(self, saved, gout):
        params = self.params()
        with _gc_paused(), torch.cuda.device(self.dev):
            sizes = [(p.numel() + 3) // 4 * 4 for p in params]
            flat = torch.zeros(sum(sizes), dtype=torch.float32, device=self.dev)
            self._grads, views, off = {}, [], 0
            for p, sz in zip(params, sizes):
                v = flat[off:off + p.numel()].view_as(p)
                self._grads[id(p)] = v
                views.append(v)
                off += sz
            self._backward(saved, gout.contiguous().float())
            self.packs.finalize()
            self._grads = None
        return views
