"""Single-launch functional wrappers over the C ABI (CUDA tensors only).

These are the building blocks the engines record into programs; they are also
what the kernel-level parity tests call.  Tensors are fp32; activations NHWC
with channel stride ``cs`` (multiple of 4, pad channels zero).
"""
import ctypes as C

import torch

from . import _lib
from .engine import pack_conv_weight, fold_scale_shift, _round_up


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise TypeError('egonet_amd.ops work on CUDA tensors only (no CPU fallback)')


def nchw_to_nhwc(x, cs=None):
    """[N,C,H,W] -> [N,H,W,cs]."""
    _need_cuda(x)
    x = x.contiguous().float()
    n, c, h, w = x.shape
    cs = _round_up(c, 4) if cs is None else cs
    y = torch.empty(n, h, w, cs, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().egn_nchw_to_nhwc_f32(_lib.ptr(x), _lib.ptr(y), n, c, h, w, cs,
                                                   _lib.current_stream(x.device)), 'nchw_to_nhwc')
    return y


def nhwc_to_nchw(x, c):
    """[N,H,W,cs] -> [N,c,H,W]."""
    _need_cuda(x)
    n, h, w, cs = x.shape
    y = torch.empty(n, c, h, w, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().egn_nhwc_to_nchw_f32(_lib.ptr(x), _lib.ptr(y), n, c, h, w, cs,
                                                   _lib.current_stream(x.device)), 'nhwc_to_nchw')
    return y


def fill_coord_ramps(y, c0):
    _need_cuda(y)
    n, h, w, cs = y.shape
    with torch.cuda.device(y.device):
        _lib.check(_lib.lib().egn_fill_coord_ramps_f32(_lib.ptr(y), n, h, w, cs, c0,
                                                       _lib.current_stream(y.device)), 'ramps')
    return y


class PackedConv(object):
    """Device-resident packed weights + folded scale/shift of one conv layer."""

    def __init__(self, weight, bias=None, bn=None, device='cuda', wino=False, kind=None):
        """``wino``: pack for the fused Winograd kernels (config kind 1) ON THE DEVICE
        (egn_wino_pack_weight_f32) instead of the direct layout; ``kind`` 2: the F(4x4,3x3) filter
        (engine.pack_wino43_weight, host float64)."""
        self.cout, self.cin, self.kh, self.kw = weight.shape
        if kind in (2, 3):
            from .engine import pack_for_kind
            self.w = pack_for_kind(weight, kind).to(device)
        elif wino or kind == 1:
            L = _lib.lib()
            nfl = L.egn_wino_weight_floats(self.cout, self.cin, 0)
            if nfl == 0 or (self.kh, self.kw) != (3, 3):
                raise ValueError('no Winograd packing for a %s filter' % (tuple(weight.shape),))
            wd = weight.detach().to(device=device, dtype=torch.float32).contiguous()
            self.w = torch.empty(nfl, dtype=torch.float32, device=device)
            with torch.cuda.device(self.w.device):
                _lib.check(L.egn_wino_pack_weight_f32(_lib.ptr(wd), self.cout, self.cin, 0, _lib.ptr(self.w),
                                                      _lib.current_stream(self.w.device)), 'wino_pack')
        else:
            self.w = pack_conv_weight(weight).to(device)
        s, b = fold_scale_shift(self.cout, bias, bn)
        self.scale, self.shift = s.to(device), b.to(device)


def conv2d_nhwc(x, pc, cin, stride=1, pad=0, act=0, res=None, out_nchw=False, cs_out=None, cfg=0):
    """x [N,H,W,cs_in] NHWC -> [N,Ho,Wo,cs_out] (or [N,Cout,Ho,Wo])."""
    _need_cuda(x, res)
    n, h, w, cs_in = x.shape
    ho = (h + 2 * pad - pc.kh) // stride + 1
    wo = (w + 2 * pad - pc.kw) // stride + 1
    if out_nchw:
        y = torch.empty(n, pc.cout, ho, wo, dtype=torch.float32, device=x.device)
        cs_o = pc.cout
    else:
        cs_o = _round_up(pc.cout, 4) if cs_out is None else cs_out
        y = torch.empty(n, ho, wo, cs_o, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().egn_conv2d_f32(
            _lib.ptr(x), _lib.ptr(pc.w), _lib.ptr(pc.scale), _lib.ptr(pc.shift), _lib.ptr(res), _lib.ptr(y),
            n, h, w, cin, cs_in, pc.cout, cs_o, pc.kh, pc.kw, stride, pad, act, int(out_nchw), cfg,
            _lib.current_stream(x.device)), 'conv2d')
    return y


def fuse_sum_relu(terms, shifts, c, relu=True):
    """terms: NHWC tensors; shifts[i] = log2 upsample factor of terms[i]."""
    _need_cuda(*terms)
    base = [t for t, s in zip(terms, shifts) if s == 0][0]
    n, h, w, cs = base.shape
    y = torch.empty_like(base)
    ptrs = (C.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
    sh = (C.c_int * len(terms))(*shifts)
    with torch.cuda.device(base.device):
        _lib.check(_lib.lib().egn_fuse_sum_relu_f32(_lib.ptr(y), n, h, w, c, cs, len(terms), ptrs, sh,
                                                    int(relu), _lib.current_stream(base.device)), 'fuse')
    return y
