"""Native training step of the FC lifter on MI355X (BASELINE config 3).

Reference hot loop: ``libs/trainer/trainer.py:183-209`` (zero_grad -> forward ->
loss -> backward -> optim.step) as driven by ``tools/train_lifting.py`` /
``trainer.train_cascade`` (trainer.py:25-71): FCModel(66 -> 96) in train mode
(BatchNorm1d on batch statistics, Dropout p), ``MSELoss1D(reduction='mean')``
(libs/loss/function.py:204-215), Adam(lr 1e-3) (libs/optimizer/optimizer.py).

``LifterTrainStep.step(x, target)`` runs forward, backward and the Adam update
as HIP launches only (C ABI, include/egonet_hip.h "training step building
blocks"); autograd is not involved.  Every GEMM -- forward ``z = a W^T``, dgrad
``da = dz W`` and wgrad ``dW = dz^T a`` -- runs on the fp32-MFMA conv kernel
with weights packed on the device each step; BatchNorm/ReLU/dropout forward and
backward, bias/column sums, the MSE and Adam are fused HBM-bound kernels
(csrc/train_ops.hip).  The module's ``nn.Parameter``s and BatchNorm buffers are
updated in place, so ``state_dict()`` / ``L.pth`` stay the interface.

Dropout uses a torch-generated keep mask (the reference's mask stream cannot be
reproduced bit for bit anyway); parity tests run with p = 0.
"""
import ctypes as C
import os

import torch

from . import _lib, tuner
from .engine import _round_up, invalidate
from .train_hrnet import FlatParams, StepCounters


class _Unit(object):
    """Linear (+ BatchNorm1d + ReLU + dropout) with everything backward needs."""

    def __init__(self, fc, bn):
        self.fc, self.bn = fc, bn
        self.inf, self.outf = fc.in_features, fc.out_features


def _rank_mixed_seed(seed):
    """seed ^ rank * golden-ratio constant, kept inside 62 bits (rank 0 / no process group: the seed itself)."""
    try:
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    except Exception:
        rank = 0
    return (seed ^ ((rank * 0x9E3779B97F4A7C15) & ((1 << 62) - 1))) & ((1 << 62) - 1)


class LifterTrainStep(object):
    # dense layers on csrc/gemm.hip where the shape allows (EGONET_AMD_GEMM=0: the conv-kernel route everywhere);
    # tile variant per form (NT, NN, TN), the fastest of tools/gemm_probe.py on 4096 x 1024 x 1024 [MI355X r3]:
    # NT 128x128 8 waves 2 stages 69 us, NN 128x128 8 waves 68 us, TN 128x128 4 waves split-K 4 74.5 us
    # dropout keep masks drawn inside the BatchNorm / ReLU kernels (EGONET_AMD_RNG_DROPOUT=0: a torch-generated mask
    # tensor per unit, the round-2 route)
    rng_dropout = os.environ.get('EGONET_AMD_RNG_DROPOUT', '1') != '0'
    _layer_epoch = 0          # Philox 'layer' word = unit + 16 * (non-updating forwards since the last optimizer step)
    use_gemm = os.environ.get('EGONET_AMD_GEMM', '1') != '0'
    # [r4] BatchNorm statistics from the forward GEMM's epilogue, skip-path gradient added in the data-gradient GEMM's
    # epilogue (EGONET_AMD_GEMM_FUSE=0: the separate column-reduction / add passes of round 3)
    fuse_gemm_epilogue = os.environ.get('EGONET_AMD_GEMM_FUSE', '1') != '0'
    _gemm_fused = False
    gemm_variant = [int(v) for v in os.environ.get('EGONET_AMD_GEMM_VARIANTS', '3,0,1').split(',')]

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, dropout=None, grad_sync=None,
                 optim_type='adam', momentum=0.0, weight_decay=0.0):
        p0 = next(model.parameters())
        if not p0.is_cuda:
            raise ValueError('LifterTrainStep needs the model on a GPU')
        if optim_type not in ('adam', 'sgd'):
            raise NotImplementedError('optimizer %r (optimizer.py:8-40 knows adam and sgd)' % (optim_type,))
        self.model = model
        self.dev = p0.device
        self.lr, self.betas, self.eps = lr, betas, eps
        self.optim_type, self.momentum, self.weight_decay = optim_type, float(momentum), float(weight_decay)
        self.act = 2 if model.leaky else 1       # nn.LeakyReLU() / nn.ReLU (FCmodel.py:19-22) in the BN kernels
        self.grad_sync = grad_sync
        self.p = float(model.p_dropout if dropout is None else dropout)
        self.units = [_Unit(model.w1, model.batch_norm1)]
        for blk in model.res_blocks:
            self.units += [_Unit(blk.w1, blk.batch_norm1), _Unit(blk.w2, blk.batch_norm2)]
        self.final = model.w2
        # parameters / gradients / Adam moments as views of flat buffers: one Adam launch,
        # one all-reduce buffer, step counter and lr on the device (hipGraph-safe)
        self.flat = FlatParams(model.parameters())
        # in-kernel dropout: the seed comes from torch's generator at construction (torch.manual_seed makes a run
        # reproducible), the per-iteration counter is the optimizer's device-resident step counter
        # Data-parallel ranks seeded alike (torch.manual_seed(s) on every rank) must not draw the same keep masks:
        # the reference's DataParallel replicas use per-device generators.  The rank is mixed into the seed.
        self.drop_seed = _rank_mixed_seed(int(torch.randint(0, 2 ** 62, (1,)).item()))
        self.drop_step = self.flat.step_dev
        # steps that do not update (update=False) leave the optimizer's counter alone: a host-side count of such
        # forwards since the last update goes into the Philox "layer" word, so that they do not repeat a mask
        self._noupdate_forwards = 0
        # every Linear weight as a 1x1 conv filter [out, in, 1, 1] (views of the flat buffer): from the second
        # step on all forward / data-gradient packs of the iteration are ONE launch (train_hrnet.PackedFilters)
        from .train_hrnet import PackedFilters
        self.packs = PackedFilters(p0.device)
        self.w4 = {}
        for fc in [u.fc for u in self.units] + [self.final]:
            self.w4[id(fc.weight)] = fc.weight.detach().view(fc.out_features, fc.in_features, 1, 1)
        self.params = self.flat.params
        self.grads = {id(p): p.grad for p in self.params}
        self._ws = {}
        self._wgrad_floats = 0
        widest = _round_up(max([u.outf for u in self.units] + [u.inf for u in self.units]
                               + [self.final.out_features]), 16) + 16
        self.ones = torch.ones(widest, dtype=torch.float32, device=self.dev)     # conv scale (no BN folding here)
        self.zeros = torch.zeros(widest, dtype=torch.float32, device=self.dev)   # conv shift for dgrad / wgrad
        self.L = _lib.lib()
        self.loss_dev = torch.zeros(1, dtype=torch.float64, device=self.dev)
        self.counters = StepCounters()
        # weight gradients on a side stream, beside the backward chain (as in train_hrnet;
        # EGONET_AMD_WGRAD_STREAM=0: one stream)
        self.wgrad_stream = torch.cuda.Stream(device=self.dev) \
            if os.environ.get('EGONET_AMD_WGRAD_STREAM', '1') != '0' else None
        self._side_used = False
        self._side_keep = []
        self.timing = None            # bench.py: a list collects (cfg, flops, start, end) per GEMM launch

    # -- helpers -----------------------------------------------------------
    def _buf(self, name, *shape):
        key = (name,) + shape
        t = self._ws.get(key)
        if t is None:
            t = torch.zeros(*shape, dtype=torch.float32, device=self.dev)
            self._ws[key] = t
        return t

    def _st(self):
        return _lib.current_stream(self.dev)

    def _stats_table(self, ui, nrow, cols):
        """float64 [nrow][2][cols] partial-statistics table of unit ``ui`` (written whole by the GEMM's epilogue)."""
        key = ('stats', ui, nrow, cols)
        t = self._ws.get(key)
        if t is None:
            t = torch.zeros(nrow * 2 * cols, dtype=torch.float64, device=self.dev)
            self._ws[key] = t
        return t

    def _gemm(self, a, rows, k, ld_a, w_src, ld_w, cout, transpose_w, out, shift=None, tagk='', addend=None, stats=None):
        """out[rows, cout] (contiguous) = a[rows, :k] @ W^T + shift, W[co][ci] taken from
        w_src (row-major, ld_w) directly (transpose_w=0) or transposed (1).  On the dense-GEMM route two epilogue
        fusions [r4]: ``addend`` ([rows, cout], data-gradient form: out = a W + addend -- the skip-path gradient of a
        residual block, no add pass) and ``stats`` (forward form: a [rows / 128][2][cout] float64 table the epilogue
        fills with partial column sums / sums of squares of ``out`` -- BatchNorm1d's batch statistics without a pass
        over z).  Returns (out, fused): ``fused`` says whether the requested fusion happened (else the caller runs
        the separate pass)."""
        L = self.L
        form = 1 if transpose_w else 0
        if self.use_gemm and L.egn_gemm_supported(form, rows, cout, k, ld_a, ld_w, cout):
            # the dense fp32-MFMA GEMM (csrc/gemm.hip): operands as they lie, no packed filter
            tm = self.timing
            if tm is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream(self.dev))
            fuse = self.fuse_gemm_epilogue
            _lib.check(L.egn_gemm_ex_f32(form, _lib.ptr(a), _lib.ptr(w_src), _lib.ptr(out),
                                         _lib.ptr(shift.detach()) if shift is not None else None,
                                         _lib.ptr(addend) if fuse and form == 1 else None,
                                         _lib.ptr(stats) if fuse and form == 0 else None,
                                         stats.numel() // (2 * cout) if fuse and form == 0 and stats is not None else 0,
                                         rows, cout, k, ld_a, ld_w, cout, self.gemm_variant[form], None, 0, self._st()),
                       'gemm')
            if tm is not None:
                e1.record(torch.cuda.current_stream(self.dev))
                tm.append((-(form + 1), 2.0 * rows * k * cout, e0, e1))
            self._gemm_fused = fuse and (addend is not None or stats is not None)
            return out
        self._gemm_fused = False
        coutp = _round_up(cout, 16)
        w4 = self.w4.get(id(w_src))
        if w4 is not None and ld_w == w4.shape[1]:
            # a Linear weight: the 1x1-conv pack (dgrad = the transposed use) -- the same layout as
            # egn_pack_matrix_f32, packed with all the others in one launch once the set is known
            wp = self.packs.get(w4, transpose_w, self._st())
        else:
            nchunk = (k + 15) // 16
            wp = self._buf('wp' + tagk, nchunk * 4 * coutp * 4)
            _lib.check(L.egn_pack_matrix_f32(_lib.ptr(w_src), ld_w, cout, k, transpose_w, _lib.ptr(wp), self._st()),
                       'pack')
        sc = self.ones
        if shift is None:
            sh = self.zeros
        else:
            if coutp == cout:                    # the bias vector already has the padded length
                sh = shift.detach()
            else:
                sh = self._buf('shift' + tagk, coutp)
                sh[:cout].copy_(shift.detach())
        # a contiguous [rows, cout] result is NHWC with cs = cout when cout % 4 == 0
        # (float4 row stores); otherwise the NCHW store path writes the same layout
        nchw = 1 if cout % 4 else 0
        key = (rows, 1, 1, k, ld_a, cout, cout, 1, 1, 1, 0, False, bool(nchw))
        cfg = tuner.choose(self.dev, key)
        tm = self.timing
        if tm is not None:         # bench.py: hipEvents around every forward / data-gradient GEMM launch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.dev))
        _lib.check(L.egn_conv2d_f32(_lib.ptr(a), _lib.ptr(wp), _lib.ptr(sc), _lib.ptr(sh), None, _lib.ptr(out),
                                    rows, 1, 1, k, ld_a, cout, cout, 1, 1, 1, 0, 0, nchw, cfg, self._st()), 'gemm')
        if tm is not None:
            e1.record(torch.cuda.current_stream(self.dev))
            tm.append((cfg, 2.0 * rows * k * cout, e0, e1))
        return out

    def _wgrad(self, a, ld_a, inf, dz, ld_dz, outf, rows, grad_w, keep=None):
        """grad_w[outf, inf] = dz^T a on the split-K MFMA weight-gradient kernel
        (both operands are read as they lie, row-major)."""
        L = self.L
        gemm = self.use_gemm and L.egn_gemm_supported(2, outf, inf, rows, ld_dz, ld_a, inf)
        need = L.egn_gemm_ws_bytes(2, outf, inf, rows) if gemm else \
            L.egn_conv2d_wgrad_ws_bytes(rows, 1, 1, inf, ld_a, outf, ld_dz, 1, 1, 1, 0)
        if need < 0:
            raise _lib.EgonetHipError('wgrad: unsupported shape')
        ws = self._buf('wgrad_ws', max(need // 4, self._wgrad_floats, 4))
        self._wgrad_floats = ws.numel()
        st = self._st()
        if self.wgrad_stream is not None:
            # a and dz are named buffers of this unit that nothing overwrites before the join
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.dev))
            self.wgrad_stream.wait_event(ev)
            st = C.c_void_p(self.wgrad_stream.cuda_stream)
            self._side_used = True
            if keep is not None:           # freshly allocated operands: alive until the side stream is joined
                self._side_keep.append(keep)
        if gemm:          # dW[out][in] = dz^T a: both operands as they lie (M-contiguous), split along the batch
            _lib.check(L.egn_gemm_f32(2, _lib.ptr(dz), _lib.ptr(a), _lib.ptr(grad_w), None, outf, inf, rows, ld_dz, ld_a,
                                      inf, self.gemm_variant[2], _lib.ptr(ws), ws.numel() * 4, st), 'wgrad gemm')
            return
        _lib.check(L.egn_conv2d_wgrad_f32(_lib.ptr(a), _lib.ptr(dz), _lib.ptr(grad_w), rows, 1, 1, inf, ld_a, outf,
                                          ld_dz, 1, 1, 1, 0, _lib.ptr(ws), ws.numel() * 4, st), 'wgrad')

    def _join_side(self):
        if self.wgrad_stream is not None and self._side_used:
            ev = torch.cuda.Event()
            ev.record(self.wgrad_stream)
            torch.cuda.current_stream(self.dev).wait_event(ev)
            self._side_used = False
        self._side_keep = []

    # -- the step -----------------------------------------------------------
    @torch.no_grad()
    def step(self, x, target, update=True):
        """One zero_grad/forward/loss/backward/Adam iteration.  Returns the loss
        (Python float is NOT forced: a 1-element float64 device tensor).  The cyclic garbage collector
        is paused while the launches are issued (train_hrnet.HRNetTrainStep.step)."""
        from .train_hrnet import _gc_paused
        with _gc_paused():
            return self._step(x, target, update)

    def grad_of(self, p):
        return self.grads[id(p)]

    def _forward(self, x, fresh=False):
        """Train-mode forward (FCmodel.py:92-105 with BatchNorm1d on batch statistics and dropout).  Returns
        (pred [B, out], saved) -- ``saved`` is what ``_backward`` needs.  ``fresh``: every activation in a new
        allocation (the autograd bridge: a second forward must not overwrite what a pending backward reads)
        instead of this object's named buffers."""
        L, dev = self.L, self.dev
        B = x.shape[0]              # any size: the reference's DataLoader has no drop_last (trainer.py:113-125)
        if B < 2:
            raise ValueError('BatchNorm1d needs more than one sample per batch in training mode')
        x = x.contiguous().float()
        st = self._st()
        keep = 1.0 / (1.0 - self.p) if self.p > 0 else 1.0

        def buf(name, *shape):
            if fresh:
                return torch.empty(*shape, dtype=torch.float32, device=dev)
            return self._buf(name, *shape)
        ws = self._buf('colws', L.egn_colreduce_ws_bytes(1024 + 16) // 4)
        self.packs.pack_all(st)           # every filter of the iteration, one launch (from the second step on)
        # input rows padded to a multiple of 4 floats
        ld0 = _round_up(self.units[0].inf, 4)
        a = buf('a0', B, ld0)
        _lib.check(L.egn_nchw_to_nhwc_f32(_lib.ptr(x), _lib.ptr(a), B, self.units[0].inf, 1, 1, ld0, st))
        saved = []
        ld_a = ld0
        block_in = None
        for ui, u in enumerate(self.units):
            z = buf('z%d' % ui, B, u.outf)
            nrow = L.egn_gemm_stats_rows(B)
            part = self._stats_table(ui, nrow, u.outf) if nrow else None
            self._gemm(a, B, u.inf, ld_a, u.fc.weight, u.inf, u.outf, 0, z, shift=u.fc.bias, tagk='f%d' % ui, stats=part)
            mean = buf('mean%d' % ui, u.outf)
            istd = buf('istd%d' % ui, u.outf)
            varu = buf('varu%d' % ui, u.outf)
            mom = u.bn.momentum if u.bn.momentum is not None else 0.1
            if part is not None and self._gemm_fused:
                # the GEMM's epilogue left partial column sums: only the finalise launch remains
                _lib.check(L.egn_bn_stats_finalize_f32(_lib.ptr(part), nrow, B, u.outf, u.bn.eps, _lib.ptr(mean),
                                                       _lib.ptr(istd), _lib.ptr(varu), _lib.ptr(u.bn.running_mean),
                                                       _lib.ptr(u.bn.running_var), mom, st), 'bn_stats_finalize')
            else:
                _lib.check(L.egn_bn_stats_f32(_lib.ptr(z), B, u.outf, u.outf, u.bn.eps, _lib.ptr(mean), _lib.ptr(istd),
                                              _lib.ptr(varu), _lib.ptr(u.bn.running_mean),
                                              _lib.ptr(u.bn.running_var), mom, _lib.ptr(ws), st), 'bn_stats')
            mask = None
            # second unit of a residual block: out = block_in + y (FCmodel.py:33-43) is written by the BatchNorm pass
            # itself (activation flag | 0x10: `res` added after activation and dropout) -- no add pass [r4]
            tail = ui > 0 and ui % 2 == 0 and self.fuse_gemm_epilogue
            y = buf(('blk%d' if tail else 'y%d') % ui, B, u.outf)
            act_flag, res_ptr = (self.act | 0x10, _lib.ptr(block_in)) if tail else (self.act, None)
            if self.p > 0 and self.rng_dropout:
                # keep mask drawn inside the kernel (Philox on (seed; element, unit, step)): the backward kernels
                # regenerate it -- no mask tensor, no RNG kernel in the step
                _lib.check(L.egn_bn_act_fwd_drop_f32(_lib.ptr(z), _lib.ptr(mean), _lib.ptr(istd), _lib.ptr(u.bn.weight),
                                                     _lib.ptr(u.bn.bias), self.p, self.drop_seed, _lib.ptr(self.drop_step),
                                                     ui + 16 * self._layer_epoch, act_flag, res_ptr, _lib.ptr(y), B, u.outf,
                                                     u.outf, st), 'bn_act_fwd_drop')
            else:
                if self.p > 0:
                    mask = buf('mask%d' % ui, B, u.outf)
                    mask.bernoulli_(1.0 - self.p)          # keep mask (0/1), one launch, capture-aware RNG
                _lib.check(L.egn_bn_act_fwd_f32(_lib.ptr(z), _lib.ptr(mean), _lib.ptr(istd), _lib.ptr(u.bn.weight),
                                                _lib.ptr(u.bn.bias), _lib.ptr(mask), keep, act_flag, res_ptr, _lib.ptr(y), B,
                                                u.outf, u.outf, st), 'bn_act_fwd')
            saved.append((a, ld_a, z, mean, istd, mask))
            if ui == 0:
                block_in = y
                a = y
            elif ui % 2 == 1:          # first unit of a residual block
                a = y
            elif tail:                 # second unit, sum already written by the BatchNorm pass
                block_in = y
                a = y
            else:                      # second unit: out = block_in + y   (FCmodel.py:33-43)
                out = buf('blk%d' % ui, B, u.outf)
                _lib.check(L.egn_add_f32(_lib.ptr(block_in), _lib.ptr(y), _lib.ptr(out), B * u.outf, st))
                block_in = out
                a = out
            ld_a = u.outf
        if not getattr(self, '_native_tick', False):       # (the native step ticks the counters with the loss reset)
            torch._foreach_add_([u.bn.num_batches_tracked for u in self.units], 1)
        feat = a
        nf = self.final.in_features
        no = self.final.out_features
        pred = buf('pred', B, no)
        self._gemm(feat, B, nf, nf, self.final.weight, nf, no, 0, pred, shift=self.final.bias, tagk='fo')
        return pred, (B, saved, feat, keep, fresh)

    def _backward(self, ctx, dpred, sess=None):
        """Backward of ``_forward`` from the gradient of the prediction; parameter gradients go to
        ``grad_of(param)`` (overwritten, not accumulated)."""
        L, dev = self.L, self.dev
        B, saved, feat, keep, fresh = ctx
        st = self._st()
        g = self.grad_of
        nf, no = self.final.in_features, self.final.out_features

        def buf(name, *shape):
            if fresh:
                return torch.empty(*shape, dtype=torch.float32, device=dev)
            return self._buf(name, *shape)
        ws = self._buf('colws', L.egn_colreduce_ws_bytes(1024 + 16) // 4)
        self._wgrad(feat, nf, nf, dpred, no, no, B, g(self.final.weight))
        _lib.check(L.egn_colsum_f32(_lib.ptr(dpred), B, no, no, _lib.ptr(g(self.final.bias)), _lib.ptr(ws), st))
        if sess is not None:
            sess.done([self.final.weight, self.final.bias])
        dy = buf('dy_top', B, nf)
        self._gemm(dpred, B, no, no, self.final.weight, nf, nf, 1, dy, tagk='do')

        d_block_out = dy              # gradient w.r.t. the output of the current residual block
        for ui in range(len(self.units) - 1, -1, -1):
            u = self.units[ui]
            a_in, ld_in, z, mean, istd, mask = saved[ui]
            if ui == 0 or ui % 2 == 0:
                d_y = d_block_out     # unit 0 and the second unit of a block see the block-output gradient
            dbeta, dgamma = g(u.bn.bias), g(u.bn.weight)
            dz = buf('dz%d' % ui, B, u.outf)     # per unit: the side stream reads it until the join
            if keep != 1.0 and mask is None:     # the forward drew its mask in the kernel: same (seed, unit, step) here
                _lib.check(L.egn_bn_bwd_sums_drop_f32(_lib.ptr(d_y), _lib.ptr(z), self.p, self.drop_seed,
                                                      _lib.ptr(self.drop_step), ui + 16 * self._layer_epoch, _lib.ptr(mean),
                                                      _lib.ptr(istd),
                                                      _lib.ptr(u.bn.weight), _lib.ptr(u.bn.bias), self.act, None, B,
                                                      u.outf, u.outf, _lib.ptr(dbeta), _lib.ptr(dgamma), _lib.ptr(ws), st),
                           'bn_bwd_sums_drop')
                _lib.check(L.egn_bn_bwd_dz_drop_f32(_lib.ptr(d_y), _lib.ptr(z), self.p, self.drop_seed,
                                                    _lib.ptr(self.drop_step), ui + 16 * self._layer_epoch, _lib.ptr(mean),
                                                      _lib.ptr(istd),
                                                    _lib.ptr(u.bn.weight), _lib.ptr(u.bn.bias), self.act, None,
                                                    _lib.ptr(dbeta), _lib.ptr(dgamma), _lib.ptr(dz), None, B, u.outf,
                                                    u.outf, st), 'bn_bwd_dz_drop')
            else:
                _lib.check(L.egn_bn_bwd_sums_f32(_lib.ptr(d_y), _lib.ptr(z), _lib.ptr(mask), keep, _lib.ptr(mean),
                                                 _lib.ptr(istd), _lib.ptr(u.bn.weight), _lib.ptr(u.bn.bias), self.act,
                                                 None, B, u.outf, u.outf, _lib.ptr(dbeta), _lib.ptr(dgamma), _lib.ptr(ws),
                                                 st), 'bn_bwd_sums')
                _lib.check(L.egn_bn_bwd_dz_f32(_lib.ptr(d_y), _lib.ptr(z), _lib.ptr(mask), keep, _lib.ptr(mean),
                                               _lib.ptr(istd), _lib.ptr(u.bn.weight), _lib.ptr(u.bn.bias), self.act, None,
                                               _lib.ptr(dbeta), _lib.ptr(dgamma), _lib.ptr(dz), None, B, u.outf,
                                               u.outf, st), 'bn_bwd_dz')
            self._wgrad(a_in, ld_in, u.inf, dz, u.outf, u.outf, B, g(u.fc.weight), keep=(a_in, dz) if fresh else None)
            _lib.check(L.egn_colsum_f32(_lib.ptr(dz), B, u.outf, u.outf, _lib.ptr(g(u.fc.bias)), _lib.ptr(ws), st))
            if sess is not None:
                sess.done([u.fc.weight, u.fc.bias, u.bn.weight, u.bn.bias])
            if ui == 0:
                break
            if ui % 2 == 0:            # second unit of a block: continue into the first unit
                da = buf('da%d' % (ui % 2), B, u.inf)
                self._gemm(dz, B, u.outf, u.outf, u.fc.weight, u.inf, u.inf, 1, da, tagk='d')
                d_y = da
            else:                      # first unit: block input gradient = skip path + branch path
                nxt = buf('dblk%d' % ((ui // 2) % 2), B, u.inf)
                self._gemm(dz, B, u.outf, u.outf, u.fc.weight, u.inf, u.inf, 1, nxt, tagk='d', addend=d_block_out)
                if not self._gemm_fused:        # conv-kernel route / fusion off: nxt holds dz W, add the skip path
                    da = buf('da%d' % (ui % 2), B, u.inf)
                    da.copy_(nxt)
                    _lib.check(L.egn_add_f32(_lib.ptr(d_block_out), _lib.ptr(da), _lib.ptr(nxt), B * u.inf, st))
                d_block_out = nxt
        self._join_side()

    def _step(self, x, target, update):
        L, dev = self.L, self.dev
        target = target.contiguous().float()
        with torch.cuda.device(dev):
            st = self._st()
            self._layer_epoch = self._noupdate_forwards        # (0 in a training loop: every step updates)
            # num_batches_tracked += 1 and loss = 0 in one launch (no ATen kernel inside the step)
            self.counters.tick([u.bn for u in self.units], self.loss_dev, st)
            self._native_tick = True
            try:
                pred, ctx = self._forward(x)
            finally:
                self._native_tick = False
            B, no = ctx[0], self.final.out_features
            # loss + gradient of the prediction
            dpred = self._buf('dpred', B, no)
            _lib.check(L.egn_mse_f32(_lib.ptr(pred), _lib.ptr(target), B, no, no, no, 1.0, 0, _lib.ptr(dpred),
                                     _lib.ptr(self.loss_dev), st), 'mse')
            sess = self.grad_sync.begin(self.flat, torch.cuda.current_stream(dev), self.wgrad_stream) \
                if hasattr(self.grad_sync, 'begin') else None
            self._backward(ctx, dpred, sess)
            if sess is not None:
                sess.finish()
            elif self.grad_sync is not None:
                self.grad_sync(self.flat.grad)
            if update:
                self.flat.update(self, st)
                self._noupdate_forwards = 0
            else:                             # the optimizer's step counter did not move: the next masks must
                self._noupdate_forwards += 1
            self.packs.finalize()             # first step: the set of filters is known now
            # weights and BatchNorm buffers were written through raw pointers: eval-mode forwards
            # between steps (eval_during, EgoNet.L after fine-tuning) must re-fold them
            invalidate(self.model)
        return self.loss_dev
