"""Launch-local checks of the native training tape (egonet_amd.train_hrnet).

End-to-end gradient comparisons of a deep ReLU network in fp32 are limited by
ReLU ties: an element whose pre-activation is within rounding of 0 gets gate 1
in one implementation and 0 in another, and everything below it changes by a
finite amount (measured on MI355X, tests/train_debug.py: HRNet-W48, 2 crops:
99 of 30.8 M block-output gates differ from torch's own GPU autograd ->
gradient cosine 0.99985; tiny net without a flipped gate: relative L2 1e-5).

So the full-size parity test checks every backward launch IN PLACE: each
weight-gradient, data-gradient and BatchNorm-backward launch of the step is
recomputed in float64 on the CPU (torch conv autograd / the BatchNorm backward
formulas) from the very tensors the launch consumed.  The composition of the
launches (the tape wiring) is pinned by the small, flip-free cases against the
reference's own iterations (tests/golden/hrnet_train.npz).
"""
import numpy as np
import torch
import torch.nn.functional as F


def rel(got, want):
    return float(np.abs(got - want).max()) / max(float(np.abs(want).max()), 1e-30)


def _nchw(t, n, h, w, cs, c, device='cpu'):
    return t.view(n, h, w, cs)[..., :c].permute(0, 3, 1, 2).double().to(device)


def wgrad_ref64(x, dy, kh, kw, stride, pad):
    """dW[co,ci,ky,kx] = sum_{n,y,x} dy[n,co,y,x] * xpad[n,ci,s*y+ky,s*x+kx] in float64, one matrix product per
    tap -- on a GPU that is rocBLAS dgemm, an implementation independent of this package's kernels (and of
    MIOpen), fast enough to check every launch of a 32-crop W48 step."""
    xp = F.pad(x, (pad, pad, pad, pad))
    n, co, ho, wo = dy.shape
    ci = x.shape[1]
    dyf = dy.permute(1, 0, 2, 3).reshape(co, -1)
    out = torch.empty(co, ci, kh, kw, dtype=torch.float64, device=x.device)
    for ky in range(kh):
        for kx in range(kw):
            xs = xp[:, :, ky:ky + stride * (ho - 1) + 1:stride, kx:kx + stride * (wo - 1) + 1:stride]
            out[:, :, ky, kx] = dyf @ xs.permute(1, 0, 2, 3).reshape(ci, -1).t()
    return out


def dgrad_ref64(dy, w, stride, pad, h, wd):
    """dx[n,ci,s*y+ky-p,s*x+kx-p] += sum_co dy[n,co,y,x] * W[co,ci,ky,kx] in float64, one matrix product per tap."""
    n, co, ho, wo = dy.shape
    ci, kh, kw = w.shape[1], w.shape[2], w.shape[3]
    dxp = torch.zeros(n, ci, h + 2 * pad, wd + 2 * pad, dtype=torch.float64, device=dy.device)
    dyf = dy.permute(0, 2, 3, 1).reshape(-1, co)
    for ky in range(kh):
        for kx in range(kw):
            t = (dyf @ w[:, :, ky, kx]).view(n, ho, wo, ci).permute(0, 3, 1, 2)
            dxp[:, :, ky:ky + stride * (ho - 1) + 1:stride, kx:kx + stride * (wo - 1) + 1:stride] += t
    return dxp[:, :, pad:pad + h, pad:pad + wd]


def _rel_t(got, want):
    return float((got - want).abs().max()) / max(float(want.abs().max()), 1e-30)


class LayerChecks(object):
    """Context manager: wraps the tape's wgrad / dgrad launches and the BatchNorm
    backward hook of an ``HRNetTrainStep``; collects relative errors."""

    def __init__(self, trainer, device='cpu'):
        """``device='cuda'``: the float64 recomputation runs on the GPU as per-tap matrix products
        (``wgrad_ref64`` / ``dgrad_ref64``: rocBLAS dgemm) instead of torch's CPU convolution autograd --
        what makes the check affordable at the bench's 32 crops."""
        self.tr = trainer
        self.device = device
        self.wgrad, self.dgrad, self.bn = [], [], []

    def __enter__(self):
        from egonet_amd import train_hrnet as T
        self._T = T
        self._orig = (T._Tape._wgrad, T._Tape._dgrad)
        orig_w, orig_d = self._orig
        me = self

        def wgrad(tape, x, xd, dy, cs_out, weight, stride, pad):
            orig_w(tape, x, xd, dy, cs_out, weight, stride, pad)
            torch.cuda.synchronize()           # the launch may be on the trainer's side stream
            cout, cin, kh, kw = weight.shape
            ho, wo = (x.h + 2 * pad - kh) // stride + 1, (x.w + 2 * pad - kw) // stride + 1
            if me.device != 'cpu':
                want = wgrad_ref64(_nchw(xd, x.n, x.h, x.w, x.cs, cin, me.device),
                                   _nchw(dy, x.n, ho, wo, cs_out, cout, me.device), kh, kw, stride, pad)
                me.wgrad.append((_rel_t(tape.o.grad_of(weight).double(), want), (x.n, x.h, x.w, cin, cout, kh, stride)))
                return
            wt = torch.zeros(cout, cin, kh, kw, dtype=torch.float64, requires_grad=True)
            with torch.enable_grad():
                F.conv2d(_nchw(xd, x.n, x.h, x.w, x.cs, cin), wt, None, stride, pad).backward(
                    _nchw(dy, x.n, ho, wo, cs_out, cout))
            me.wgrad.append((rel(tape.o.grad_of(weight).double().cpu().numpy(), wt.grad.numpy()),
                             (x.n, x.h, x.w, cin, cout, kh, stride)))

        def dgrad(tape, dy, ho, wo, cs_out, weight, stride, pad, x, into=None):
            # ``into``: the conv adds the gradient x already has in its epilogue and writes in place
            before = None if into is None else into.clone()
            dx = orig_d(tape, dy, ho, wo, cs_out, weight, stride, pad, x, into=into)
            if before is not None:
                assert dx.data_ptr() == into.data_ptr()
                dx = dx - before
            cout, cin, kh, kw = weight.shape
            if me.device != 'cpu':
                want = dgrad_ref64(_nchw(dy, x.n, ho, wo, cs_out, cout, me.device),
                                   weight.detach().double().to(me.device), stride, pad, x.h, x.w)
                e = _rel_t(_nchw(dx, x.n, x.h, x.w, x.cs, cin, me.device), want)
                padmax = float(dx.view(x.n, x.h, x.w, x.cs)[..., cin:].abs().max()) if x.cs > cin else 0.0
                me.dgrad.append((max(e, padmax), (x.n, x.h, x.w, cin, cout, kh, stride)))
                return into if into is not None else dx
            xs = torch.zeros(x.n, cin, x.h, x.w, dtype=torch.float64, requires_grad=True)
            with torch.enable_grad():
                F.conv2d(xs, weight.detach().double().cpu(), None, stride, pad).backward(
                    _nchw(dy, x.n, ho, wo, cs_out, cout))
            e = rel(_nchw(dx, x.n, x.h, x.w, x.cs, cin).numpy(), xs.grad.numpy())
            padmax = float(dx.view(x.n, x.h, x.w, x.cs)[..., cin:].abs().max()) if x.cs > cin else 0.0
            me.dgrad.append((max(e, padmax), (x.n, x.h, x.w, cin, cout, kh, stride)))
            return into if into is not None else dx

        T._Tape._wgrad, T._Tape._dgrad = wgrad, dgrad
        self._prev_hook = self.tr.debug_hook
        self.tr.debug_hook = self._bn_hook
        return self

    def __exit__(self, *exc):
        self._T._Tape._wgrad, self._T._Tape._dgrad = self._orig
        self.tr.debug_hook = self._prev_hook
        return False

    def _bn_hook(self, d):
        rows, cols, ld = d['rows'], d['cols'], d['ld']

        dev = self.device

        def v(t):
            return t.view(rows, ld)[:, :cols].double().to(dev)
        z, dy = v(d['z']), v(d['dy'])
        mean, istd = d['mean'].double().to(dev), d['istd'].double().to(dev)
        gm, bt = d['bn'].weight.detach().double().to(dev), d['bn'].bias.detach().double().to(dev)
        xhat = (z - mean) * istd
        res = v(d['res']) if d['res'] is not None else None
        pre = gm * xhat + bt + (res if res is not None else 0)
        # ReLU ties: a pre-activation within fp32 rounding of 0 (the kernel forms it in fp32, this check in
        # float64) gets gate 1 in one and 0 in the other -- at 32 crops a layer has millions of elements and a
        # few of them tie.  Those elements are compared with EITHER gate; what their |dy| can move the column
        # sums by is added to the sums' bound.
        tie = None
        if d['relu']:
            mag = (gm * xhat).abs() + bt.abs() + (res.abs() if res is not None else 0)
            tie = pre.abs() <= 4e-7 * mag
        dpre = dy * (pre > 0) if d['relu'] else dy
        dbeta, dgamma = dpre.sum(0), (dpre * xhat).sum(0)
        dz = gm * istd * (dpre - dbeta / rows - xhat * dgamma / rows)

        def elem(got, want, alt=None):
            diff = (got - want).abs()
            if tie is not None and alt is not None:
                diff = torch.where(tie, torch.minimum(diff, (got - alt).abs()), diff)
            return float(diff.max()) / max(float(want.abs().max()), 1e-30)
        slack_b = slack_g = 0.0
        dz_alt = dres_alt = None
        if tie is not None and bool(tie.any()):
            slack_b = float((dy.abs() * tie).sum(0).max()) / max(float(dbeta.abs().max()), 1e-30)
            slack_g = float(((dy * xhat).abs() * tie).sum(0).max()) / max(float(dgamma.abs().max()), 1e-30)
            dpre_alt = torch.where(tie, dy - dpre, dpre)            # the other gate on the tied elements
            dz_alt = gm * istd * (dpre_alt - dbeta / rows - xhat * dgamma / rows)
            dres_alt = dpre_alt
        self.ties = getattr(self, 'ties', 0) + (int(tie.sum()) if tie is not None else 0)
        errs = [max(elem(v(d['dz']), dz, dz_alt) - (slack_b + slack_g), 0.0),
                max(_rel_t(d['dbeta'].double().to(dev), dbeta) - slack_b, 0.0),
                max(_rel_t(d['dgamma'].double().to(dev), dgamma) - slack_g, 0.0),
                elem(v(d['dres']), dpre, dres_alt) if d['dres'] is not None else 0.0,
                _rel_t(mean, z.mean(0)),
                _rel_t(istd, (z.var(0, unbiased=False) + d['bn'].eps).rsqrt())]
        self.bn.append((max(errs), d['tag'], (rows, cols, ld), errs))
        if self._prev_hook is not None:
            self._prev_hook(d)

    def worst(self):
        return {k: max([r[0] for r in getattr(self, k)] or [0.0]) for k in ('wgrad', 'dgrad', 'bn')}


def gradient_agreement(named_params, oracle_grads):
    """(global relative L2, cosine, median per-tensor relative L2) of the module's
    .grad tensors against a dict of oracle gradients."""
    a = np.concatenate([named_params[k].grad.cpu().numpy().ravel().astype(np.float64) for k in oracle_grads])
    b = np.concatenate([oracle_grads[k].numpy().ravel().astype(np.float64) for k in oracle_grads])
    per = [np.linalg.norm(named_params[k].grad.cpu().numpy().astype(np.float64) - oracle_grads[k].numpy())
           / max(np.linalg.norm(oracle_grads[k].numpy().astype(np.float64)), 1e-30) for k in oracle_grads]
    return (float(np.linalg.norm(a - b) / np.linalg.norm(b)),
            float(a @ b / np.linalg.norm(a) / np.linalg.norm(b)), float(np.median(per)))
