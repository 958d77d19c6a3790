"""Execution engines: turn a parameter-owning ``nn.Module`` into a native launch
program of gfx950 HIP kernels (C ABI: include/egonet_hip.h).

``HRNetEngine``   backbone + heads        (reference hrnet.py:563-614)
``LifterEngine``  2D->3D residual MLP      (reference FCmodel.py:92-105)

A program is recorded once per (device, batch shape):
  * ``_Recorder`` walks the module tree and records symbolic ops on symbolic
    NHWC buffers (channel stride rounded up to 4);
  * buffer lifetimes are derived from the op list and packed into ONE activation
    arena by a best-fit free-list allocator (an HRNet-W48 forward at B=64 needs
    a few hundred MB instead of ~18 GB of distinct outputs, so the working set
    of neighbouring layers stays inside the 256 MB Infinity Cache);
  * weights are folded (BatchNorm -> per-channel scale/shift) and prepacked into
    ONE device blob in the layout the conv kernel stages through LDS;
  * the ops are emitted into a native ``egn_program`` whose pointers are
    (slot, offset) pairs: slot 0 arena, 1 weights, 2.. user tensors.
A forward is then a single C call that issues ~320 kernel launches.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import Ref, NULL_REF, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_LEAKY

CK = 16                 # input channels per K chunk (EGN_CK in csrc/egn_internal.h)
ACT_RES_AFTER = 0x10
BN_EPS_DEFAULT = 1e-5
SLOT_ARENA, SLOT_WEIGHTS, SLOT_USER0 = 0, 1, 2


def _round_up(v, m):
    return (v + m - 1) // m * m


# ---------------------------------------------------------------------------
# weight packing
# ---------------------------------------------------------------------------
def pack_conv_weight(w):
    """[Cout,Cin,KH,KW] -> flat fp32 [nchunk][KH*KW][CK/4][CoutP][4]
    (ci = chunk*16 + quad*4 + r; zero padded)."""
    w = w.detach().to(torch.float32).cpu()
    cout, cin, kh, kw = w.shape
    coutp = _round_up(cout, 16)
    nchunk = (cin + CK - 1) // CK
    wp = torch.zeros(coutp, nchunk * CK, kh * kw, dtype=torch.float32)
    wp[:cout, :cin] = w.reshape(cout, cin, kh * kw)
    wp = wp.view(coutp, nchunk, CK // 4, 4, kh * kw).permute(1, 4, 2, 0, 3).contiguous()
    return wp.reshape(-1)


def wino_cot(cout):
    """Output channels per block (co-tile) of the Winograd kernels: 48 where Cout allows (the W48
    widths), else 32 (the W32 / Pedestrian widths and 64-channel layers); 0 = not supported
    (egn_wino_cot in csrc/egn_internal.h)."""
    return 48 if cout % 48 == 0 else (32 if cout % 32 == 0 else 0)


def pack_wino_weight(w):
    """[Cout,Cin,3,3] -> the Winograd F(2x2,3x3) filter U = G g G^T (float64, rounded once to
    fp32) in the layout conv_wino.hip stages through LDS:
    flat fp32 [co-tile = Cout/T][chunk = Cin/16][f = 4i+j][quad][T][4], T = wino_cot(Cout)
    (ci = chunk*16 + quad*4 + r), one contiguous slab (48 KB for T = 48) per (co-tile, chunk)."""
    w = w.detach().to(torch.float64).cpu()
    cout, cin, kh, kw = w.shape
    cot = wino_cot(cout)
    assert (kh, kw) == (3, 3) and cot and cin % CK == 0, w.shape
    G = torch.tensor([[1., 0., 0.], [.5, .5, .5], [.5, -.5, .5], [0., 0., 1.]], dtype=torch.float64)
    u = torch.einsum('ia,ocab,jb->ocij', G, w, G).to(torch.float32)        # [Cout,Cin,4,4]
    u = u.reshape(cout // cot, cot, cin // CK, CK // 4, 4, 16)             # ct, co, chunk, quad, r, f
    return u.permute(0, 2, 5, 3, 1, 4).contiguous().reshape(-1)


def pack_wino43_weight(w):
    """[Cout,Cin,3,3] -> the Winograd F(4x4,3x3) filter U = G g G^T (points 0, +-1, +-2; float64, rounded once to
    fp32) in the layout conv_wino43_kernel stages through LDS:
    flat fp32 [co-tile = Cout/48][K step = Cin/4][f = 6i+j][kq = ci % 4][48], one contiguous 27 KB slab per
    (co-tile, K step)."""
    w = w.detach().to(torch.float64).cpu()
    cout, cin, kh, kw = w.shape
    assert (kh, kw) == (3, 3) and cout % 48 == 0 and cin % 4 == 0, w.shape
    G = torch.tensor([[1 / 4., 0., 0.], [-1 / 6., -1 / 6., -1 / 6.], [-1 / 6., 1 / 6., -1 / 6.],
                      [1 / 24., 1 / 12., 1 / 6.], [1 / 24., -1 / 12., 1 / 6.], [0., 0., 1.]], dtype=torch.float64)
    u = torch.einsum('ia,ocab,jb->ocij', G, w, G).to(torch.float32)        # [Cout,Cin,6,6]
    u = u.reshape(cout // 48, 48, cin // 4, 4, 36)                          # ct, col, step, kq, f
    return u.permute(0, 2, 4, 3, 1).contiguous().reshape(-1)


def pack_wino4_weight(w):
    """[Cout,Cin,3,3] -> the Winograd F(4x4,3x3) filter U = G g G^T (float64, rounded once to fp32) in the layout
    conv_wino4_kernel's waves load straight into MFMA B registers (csrc/conv_wino4.hip), three dwordx4 per wave
    and k-group: flat fp32 [co-tile = Cout/48][stage = Cin/8][k-group g][wave 0..11][q 0..2][lane = 16 kq + li][4]
    where value p = 4 q + r (p = 3 pl + nt < 9, the rest is padding) is point 3 wave + pl, co = 48 ct + 16 nt + li,
    ci = 8 stage + 4 g + kq."""
    w = w.detach().to(torch.float64).cpu()
    cout, cin, kh, kw = w.shape
    assert (kh, kw) == (3, 3) and cout % 48 == 0 and cin % 8 == 0, w.shape
    G = torch.tensor([[1 / 4., 0., 0.], [-1 / 6., -1 / 6., -1 / 6.], [-1 / 6., 1 / 6., -1 / 6.],
                      [1 / 24., 1 / 12., 1 / 6.], [1 / 24., -1 / 12., 1 / 6.], [0., 0., 1.]], dtype=torch.float64)
    u = torch.einsum('ia,ocab,jb->ocij', G, w, G).to(torch.float32).reshape(cout, cin, 36)
    #   co = (ct, nt, li)          ci = (stage, g, kq)        pt = (wave, pl)
    u = u.reshape(cout // 48, 3, 16, cin // 8, 2, 4, 12, 3)   # ct nt li stage g kq wave pl
    u = u.permute(0, 3, 4, 6, 7, 1, 5, 2).contiguous()        # ct stage g wave pl nt kq li
    ct, st = cout // 48, cin // 8
    u = u.reshape(ct, st, 2, 12, 9, 64)                        # ... p = 3 pl + nt, lane = 16 kq + li
    pad = torch.zeros(ct, st, 2, 12, 12, 64, dtype=torch.float32)
    pad[:, :, :, :, :9] = u
    return pad.reshape(ct, st, 2, 12, 3, 4, 64).permute(0, 1, 2, 3, 4, 6, 5).contiguous().reshape(-1)   # q lane r


def pack_for_kind(w, kind):
    """The filter in the layout the kernels of a tile-configuration KIND read (egn_conv_config_kind):
    0 direct, 1 Winograd F(2x2,3x3), 2 / 3 Winograd F(4x4,3x3) (conv_wino43_kernel / conv_wino4_kernel)."""
    if kind == 3:
        return pack_wino4_weight(w)
    return pack_wino43_weight(w) if kind == 2 else (pack_wino_weight(w) if kind == 1 else pack_conv_weight(w))


def fold_scale_shift(cout, bias=None, bn=None):
    """Per-channel (scale, shift) so that  bn(conv(x) + bias) == conv(x)*scale + shift,
    eval-mode BatchNorm (running stats).  Computed in float64, stored fp32,
    zero padded to CoutP."""
    coutp = _round_up(cout, 16)
    scale = torch.ones(cout, dtype=torch.float64)
    shift = torch.zeros(cout, dtype=torch.float64)
    if bias is not None:
        shift = bias.detach().double().cpu().clone()
    if bn is not None:
        g = bn.weight.detach().double().cpu() if bn.weight is not None else torch.ones(cout, dtype=torch.float64)
        b = bn.bias.detach().double().cpu() if bn.bias is not None else torch.zeros(cout, dtype=torch.float64)
        inv = 1.0 / torch.sqrt(bn.running_var.detach().double().cpu() + bn.eps)
        s = g * inv
        shift = b + (shift - bn.running_mean.detach().double().cpu()) * s
        scale = s
    out_s = torch.zeros(coutp, dtype=torch.float32)
    out_b = torch.zeros(coutp, dtype=torch.float32)
    out_s[:cout] = scale.float()
    out_b[:cout] = shift.float()
    return out_s, out_b


def fold_pw(weight, bn):
    """A 1x1 filter [Cout,Cin,1,1] + eval-mode BatchNorm for csrc/conv_pw.hip: (W' [Cout*Cin] = scale * W, shift [Cout]),
    float64 arithmetic, one rounding to fp32 -- the kernel's lanes read float4 pieces of the row-major matrix as it lies."""
    cout, cin = weight.shape[:2]
    scale, shift = fold_scale_shift(cout, None, bn)
    w = weight.detach().double().cpu().reshape(cout, cin) * scale[:cout].double()[:, None]
    return w.float().reshape(-1).contiguous(), shift[:cout].clone()


# ---------------------------------------------------------------------------
# symbolic recording
# ---------------------------------------------------------------------------
class Buf(object):
    """Symbolic fp32 tensor.  NHWC [n,h,w,cs] (cs = channel stride) in the arena,
    or an external user tensor bound to ``slot``."""
    __slots__ = ('n', 'h', 'w', 'c', 'cs', 'slot', 'off', 'nbytes', 'first', 'last', 'uses', 'name')

    def __init__(self, n, h, w, c, cs=None, slot=SLOT_ARENA, nbytes=None, name=''):
        self.n, self.h, self.w, self.c = n, h, w, c
        self.cs = _round_up(c, 4) if cs is None else cs
        self.slot = slot
        self.off = 0
        self.nbytes = nbytes if nbytes is not None else n * h * w * self.cs * 4
        self.first = self.last = -1
        self.uses = []          # indices of every op that reads or writes the buffer
        self.name = name

    def ref(self):
        return Ref(self.slot, self.off)


class _Recorder(object):
    def __init__(self):
        self.ops = []          # (kind, dict)
        self.blobs = []        # list of 1-D fp32 tensors -> weights blob
        self.blob_off = 0
        self.bufs = []
        self.region = 0        # fork/join regions are totally ordered
        self.cur_lane = 0      # launch lane inside the current region

    def _push(self, kind, op):
        op['region'], op['lane'] = self.region, self.cur_lane
        self.ops.append((kind, op))

    def fork(self):
        """Ops recorded until ``join`` may run concurrently when they are on
        different lanes (``lane(k)``); they must be independent."""
        self.region += 1
        self.cur_lane = 0
        self._push('fork', {})

    def lane(self, k):
        self.cur_lane = k % 4

    def join(self):
        self.cur_lane = 0
        self._push('join', {})
        self.region += 1

    def happens_before(self, u, d):
        """Op u is complete before op d starts: earlier region, or same region
        and same lane (stream order)."""
        (ru, lu), (rd, ld) = [(self.ops[i][1]['region'], self.ops[i][1]['lane']) for i in (u, d)]
        return ru < rd or (ru == rd and lu == ld and u < d)

    # -- storage --
    def new(self, n, h, w, c, name=''):
        b = Buf(n, h, w, c, name=name)
        self.bufs.append(b)
        return b

    def weight(self, t):
        off = self.blob_off
        self.blobs.append(t)
        self.blob_off += _round_up(t.numel() * 4, 256)
        return Ref(SLOT_WEIGHTS, off)

    def _touch(self, *bufs):
        i = len(self.ops)
        for b in bufs:
            if b is None:
                continue
            if b.first < 0:
                b.first = i
            b.last = i
            b.uses.append(i)

    # -- ops --
    def conv(self, x, weight, bias=None, bn=None, act=ACT_NONE, res=None, stride=1, pad=0,
             dst=None, out_nchw=False, cout_cs=None, tag=''):
        cout, cin, kh, kw = weight.shape
        assert cin == x.c, (cin, x.c, tag)
        ho = (x.h + 2 * pad - kh) // stride + 1
        wo = (x.w + 2 * pad - kw) // stride + 1
        if dst is None:
            dst = self.new(x.n, ho, wo, cout, name=tag)
            if cout_cs is not None:
                dst.cs = cout_cs
                dst.nbytes = x.n * ho * wo * cout_cs * 4
        scale, shift = fold_scale_shift(cout, bias, bn)
        # the filter is packed when the program is built: the layout depends on the tile
        # configuration the tuner picks for the shape (direct vs Winograd kernels)
        op = dict(x=x, w=None, w_src=weight, scale=self.weight(scale),
                  shift=self.weight(shift), res=res, y=dst, cin=cin, cout=cout, kh=kh, kw=kw,
                  stride=stride, pad=pad, act=act, out_nchw=int(out_nchw), ho=ho, wo=wo, tag=tag)
        self._touch(x, res, dst)
        self._push('conv', op)
        return dst

    def fuse(self, terms, relu, tag=''):
        """terms: list of (Buf, shift); output resolution = that of a shift-0 term."""
        base = [t for t, s in terms if s == 0][0]
        y = self.new(base.n, base.h, base.w, base.c, name=tag)
        self._touch(y, *[t for t, _ in terms])
        self._push('fuse', dict(y=y, terms=terms, relu=int(relu), tag=tag))
        return y

    def pw_pair(self, h, conv3, bn3, res, relu1, conv1=None, bn1=None, tag=''):
        """layer1's 1x1 pair as one launch (csrc/conv_pw.hip): out = act(bn3(conv3(h)) (+ res)) [64 -> 256] and, with
        conv1, hn = relu(bn1(conv1(out))) [256 -> 64].  Returns (out, hn or None)."""
        assert tuple(conv3.weight.shape) == (256, 64, 1, 1) and h.c == 64 and h.cs == 64, (conv3.weight.shape, h.c, h.cs)
        out = self.new(h.n, h.h, h.w, 256, name=tag + '.out')
        w3, s3 = fold_pw(conv3.weight, bn3)
        op = dict(h=h, res=res, out=out, hn=None, w3=self.weight(w3), shift3=self.weight(s3), w1=None, shift1=None,
                  relu1=int(relu1), m=h.n * h.h * h.w, tag=tag)
        hn = None
        if conv1 is not None:
            assert tuple(conv1.weight.shape) == (64, 256, 1, 1), conv1.weight.shape
            hn = self.new(h.n, h.h, h.w, 64, name=tag + '.next')
            w1, s1 = fold_pw(conv1.weight, bn1)
            op.update(hn=hn, w1=self.weight(w1), shift1=self.weight(s1))
        self._touch(h, res, out, hn)
        self._push('pwpair', op)
        return out, hn

    def nchw_to_nhwc(self, x_ext, n, c, h, w, tag=''):
        y = self.new(n, h, w, c, name=tag)
        self._touch(y)
        self._push('to_nhwc', dict(x=x_ext, y=y, tag=tag))
        return y

    def nhwc_to_nchw(self, x, c, dst_ext, tag=''):
        self._touch(x)
        self._push('to_nchw', dict(x=x, y=dst_ext, c=c, tag=tag))

    def pixel_shuffle(self, x, c, up, dst_ext, tag=''):
        self._touch(x)
        self._push('pixshuf', dict(x=x, y=dst_ext, c=c, up=up, tag=tag))

    def ramps(self, y, c0, tag=''):
        self._touch(y)
        self._push('ramps', dict(y=y, c0=c0, tag=tag))

    def decode(self, hm_ext, n, k, h, w, mode, xy, mx, idx, tag=''):
        self._push('decode', dict(hm=hm_ext, n=n, k=k, h=h, w=w, mode=mode, xy=xy, mx=mx, idx=idx, tag=tag))

    # -- finalisation --
    def plan_arena(self):
        """Greedy best-fit packing of arena buffers by lifetime; returns bytes.
        A buffer's storage is re-used only by a buffer whose defining op starts
        after EVERY use of the old one has completed (``happens_before``), so
        concurrent lanes never alias each other's live tensors."""
        live = [b for b in self.bufs if b.slot == SLOT_ARENA and b.first >= 0]
        events = sorted(live, key=lambda b: b.first)
        free = []              # [off, size, uses]: uses = op indices that touched the bytes
        active = []
        top = 0
        for b in events:
            d = b.first
            region_d = self.ops[d][1]['region']
            # retire buffers with no later use: their bytes become candidates, tagged
            # with the uses a new owner has to wait for
            still = []
            for ab in active:
                if ab.last < d:
                    free.append([ab.off, _round_up(ab.nbytes, 256), list(ab.uses)])
                else:
                    still.append(ab)
            active = still
            # uses in earlier regions are ordered before everything from here on
            for blk in free:
                blk[2] = [u for u in blk[2] if self.ops[u][1]['region'] >= region_d]
            free.sort(key=lambda blk: blk[0])
            merged = []
            for off, sz, uses in free:
                if merged and merged[-1][0] + merged[-1][1] == off:
                    merged[-1][1] += sz
                    merged[-1][2] = merged[-1][2] + uses
                else:
                    merged.append([off, sz, uses])
            free = merged

            def usable(blk):
                return all(self.happens_before(u, d) for u in blk[2])

            need = _round_up(b.nbytes, 256)
            best = None
            for i, blk in enumerate(free):
                if blk[1] >= need and usable(blk) and (best is None or blk[1] < free[best][1]):
                    best = i
            if best is None:
                # grow: extend a usable free block that ends at the top if there is one
                if free and free[-1][0] + free[-1][1] == top and usable(free[-1]):
                    off, sz, uses = free.pop()
                    b.off = off
                    top = off + need
                else:
                    b.off = top
                    top += need
            else:
                off, sz, uses = free.pop(best)
                b.off = off
                if sz > need:
                    free.append([off + need, sz - need, uses])
            active.append(b)
        return top

    def pack_pending(self):
        """Filters no tile configuration was chosen for get the direct layout."""
        for kind, op in self.ops:
            if kind == 'conv' and op.get('w') is None:
                op['w'] = self.weight(pack_conv_weight(op['w_src']))

    def weights_blob(self, device):
        self.pack_pending()
        blob = torch.zeros(max(self.blob_off, 256) // 4, dtype=torch.float32)
        off = 0
        for t in self.blobs:
            blob[off // 4: off // 4 + t.numel()] = t
            off += _round_up(t.numel() * 4, 256)
        return blob.to(device)


class Program(object):
    """Owns a native egn_program plus the torch storage its slots point to."""

    def __init__(self, rec, device, n_user_slots):
        L = _lib.lib()
        self.lib = L
        self.device = device
        arena_bytes = rec.plan_arena()
        self.arena = torch.zeros(max(arena_bytes, 256) // 4, dtype=torch.float32, device=device)
        self._choose_and_pack(rec, device)
        self.weights = rec.weights_blob(device)
        self.handle = C.c_void_p(L.egn_program_create(SLOT_USER0 + n_user_slots))
        if not self.handle:
            raise _lib.EgonetHipError('egn_program_create failed')
        self.arena_bytes = arena_bytes
        self.weight_bytes = rec.blob_off
        self.tags = []
        self.meta = []
        _lib.check(L.egn_program_bind(self.handle, SLOT_ARENA, _lib.ptr(self.arena)))
        _lib.check(L.egn_program_bind(self.handle, SLOT_WEIGHTS, _lib.ptr(self.weights)))
        # (ops may own device memory -- the ticket words of K-split convolutions: allocate it on the program's device)
        with torch.cuda.device(device):
            for kind, op in rec.ops:
                self._emit(kind, op)

    @staticmethod
    def _conv_key(op):
        x, y = op['x'], op['y']
        cs_out = y.cs if not op['out_nchw'] else op['cout']
        return (x.n, x.h, x.w, op['cin'], x.cs, op['cout'], cs_out, op['kh'], op['kw'], op['stride'],
                op['pad'], op['res'] is not None, bool(op['out_nchw']))

    def _choose_and_pack(self, rec, device):
        """Tile configuration per conv (measured table / autotune) and the filter in the layout
        that configuration's kernel stages through LDS."""
        from . import tuner
        L = self.lib
        for kind, op in rec.ops:
            if kind != 'conv' or op.get('w') is not None:
                continue
            if 'cfg' not in op:
                plain_act = (op['act'] & 0xf) in (ACT_NONE, ACT_RELU) and not (op['act'] & ACT_RES_AFTER)
                op['cfg'] = tuner.choose(device, self._conv_key(op), allow_wino=plain_act, allow_f43=plain_act)
            kind = L.egn_conv_config_kind(op['cfg']) if op['cfg'] > 0 else 0
            op['w'] = rec.weight(pack_for_kind(op['w_src'], kind))

    def _emit(self, kind, op):
        L, h = self.lib, self.handle
        flops, nbytes = 0.0, 0.0
        if kind in ('fork', 'join'):
            _lib.check(L.egn_program_fork(h) if kind == 'fork' else L.egn_program_join(h))
            self.meta.append(dict(kind=kind, klass=kind, tag='', flops=0.0, bytes=0.0, cfg=0))
            return
        _lib.check(L.egn_program_set_lane(h, op.get('lane', 0)))
        if kind == 'conv':
            x, y = op['x'], op['y']
            res = op['res'].ref() if op['res'] is not None else NULL_REF
            cs_out = y.cs if not op['out_nchw'] else op['cout']
            _lib.check(L.egn_program_add_conv2d(
                h, x.ref(), op['w'], op['scale'], op['shift'], res, y.ref(),
                x.n, x.h, x.w, op['cin'], x.cs, op['cout'], cs_out, op['kh'], op['kw'],
                op['stride'], op['pad'], op['act'], op['out_nchw'], op.get('cfg', 0)), op['tag'])
            m = x.n * op['ho'] * op['wo']
            flops = 2.0 * m * op['cout'] * op['cin'] * op['kh'] * op['kw']
            nbytes = 4.0 * (x.n * x.h * x.w * op['cin'] + m * op['cout'] * (2 if op['res'] is not None else 1)
                            + op['cout'] * op['cin'] * op['kh'] * op['kw'])
            klass = 'conv%dx%ds%d %d->%d@%dx%d' % (op['kh'], op['kw'], op['stride'], op['cin'],
                                                    op['cout'], op['ho'], op['wo'])
        elif kind == 'fuse':
            y = op['y']
            terms = op['terms']
            refs = (Ref * len(terms))(*[t.ref() for t, _ in terms])
            shifts = (C.c_int * len(terms))(*[s for _, s in terms])
            _lib.check(L.egn_program_add_fuse(h, y.ref(), y.n, y.h, y.w, y.c, y.cs, len(terms), refs,
                                              shifts, op['relu']), op['tag'])
            nbytes = 4.0 * y.n * y.c * (y.h * y.w + sum((y.h >> s) * (y.w >> s) for _, s in terms))
            klass = 'fuse%d %d@%dx%d' % (len(terms), y.c, y.h, y.w)
        elif kind == 'to_nhwc':
            y = op['y']
            _lib.check(L.egn_program_add_nchw_to_nhwc(h, op['x'], y.ref(), y.n, y.c, y.h, y.w, y.cs))
            nbytes = 4.0 * y.n * y.h * y.w * (y.c + y.cs)
            klass = 'nchw2nhwc %d@%dx%d' % (y.c, y.h, y.w)
        elif kind == 'to_nchw':
            x = op['x']
            _lib.check(L.egn_program_add_nhwc_to_nchw(h, x.ref(), op['y'], x.n, op['c'], x.h, x.w, x.cs))
            nbytes = 8.0 * x.n * x.h * x.w * op['c']
            klass = 'nhwc2nchw %d@%dx%d' % (op['c'], x.h, x.w)
        elif kind == 'pixshuf':
            x = op['x']
            _lib.check(L.egn_program_add_pixel_shuffle(h, x.ref(), op['y'], x.n, op['c'], x.h, x.w, x.cs, op['up']))
            nbytes = 8.0 * x.n * x.h * x.w * op['c'] * op['up'] ** 2
            klass = 'pixel_shuffle%d %d@%dx%d' % (op['up'], op['c'], x.h, x.w)
        elif kind == 'pwpair':
            hb, rb, ob, nb = op['h'], op['res'], op['out'], op['hn']
            none = NULL_REF
            _lib.check(L.egn_program_add_pw_pair(h, hb.ref(), rb.ref() if rb is not None else none, op['w3'], op['shift3'],
                                                 op['w1'] if nb is not None else none,
                                                 op['shift1'] if nb is not None else none, ob.ref(),
                                                 nb.ref() if nb is not None else none, op['m'], op['relu1']))
            flops = 2.0 * op['m'] * 64 * 256 * (2 if nb is not None else 1)
            nbytes = 4.0 * op['m'] * (64 + 256 * (2 if rb is not None else 1) + (64 if nb is not None else 0))
            klass = 'pwpair%d%d 64->256%s@%dx%d' % (int(rb is not None), op['relu1'], '->64' if nb is not None else '',
                                                    hb.h, hb.w)
        elif kind == 'ramps':
            y = op['y']
            _lib.check(L.egn_program_add_ramps(h, y.ref(), y.n, y.h, y.w, y.cs, op['c0']))
            nbytes = 8.0 * y.n * y.h * y.w
            klass = 'ramps'
        elif kind == 'decode':
            _lib.check(L.egn_program_add_decode(h, op['hm'], op['n'], op['k'], op['h'], op['w'],
                                                op['mode'], op['xy'], op['mx'], op['idx']))
            nbytes = 4.0 * op['n'] * op['k'] * op['h'] * op['w']
            klass = 'decode%d %dx%d' % (op['mode'], op['h'], op['w'])
        else:
            raise ValueError(kind)
        L.egn_program_tag(h, (op.get('tag') or klass).encode(), flops, nbytes)
        self.meta.append(dict(kind=kind, klass=klass, tag=op.get('tag', ''), flops=flops, bytes=nbytes,
                              cfg=op.get('cfg', 0) if kind == 'conv' else 0))

    def bind(self, slot, tensor):
        _lib.check(self.lib.egn_program_bind(self.handle, slot, _lib.ptr(tensor)))

    def run(self):
        _lib.check(self.lib.egn_program_run(self.handle, _lib.current_stream(self.device)), 'program_run')

    def run_timed(self):
        n = self.lib.egn_program_num_ops(self.handle)
        ms = (C.c_float * n)()
        _lib.check(self.lib.egn_program_run_timed(self.handle, _lib.current_stream(self.device), ms, n))
        return np.array(ms[:], dtype=np.float64)

    def capture(self):
        _lib.check(self.lib.egn_program_capture(self.handle, _lib.current_stream(self.device)))

    def replay(self):
        _lib.check(self.lib.egn_program_replay(self.handle, _lib.current_stream(self.device)))

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.egn_program_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


# ---------------------------------------------------------------------------
# staleness of the packed weights
# ---------------------------------------------------------------------------
class _Stamp(object):
    """Detects that a module's parameters / buffers changed since its programs were built.

    The packed blob folds EVERY parameter and BatchNorm buffer, so all of them are watched:
    the sum of their autograd version counters (optimizer steps, load_state_dict, in-place
    edits, BatchNorm running-stat updates), the XOR of their storage addresses (``.data = ...``,
    ``.to()``), their count, and a generation counter that code writing through raw pointers
    (the native training steps: their Adam / BatchNorm kernels never touch ``_version``) bumps
    with ``invalidate``.  ~0.2 ms on the host for the 1 828 tensors of HRNet-W48, hidden behind
    the asynchronous launches of the forward."""

    def __init__(self, model):
        self.model = model
        self.tensors = None
        self.last = None

    def _now(self):
        if self.tensors is None:
            self.tensors = list(self.model.parameters()) + list(self.model.buffers())
        ver, addr = 0, 0
        for t in self.tensors:
            ver += t._version
            addr ^= t.data_ptr()
        return (ver, addr, len(self.tensors), getattr(self.model, '_egn_generation', 0),
                self.tensors[0].device if self.tensors else None)

    def changed(self):
        now = self._now()
        if now != self.last:
            self.tensors = None          # something moved: enumerate the tensors afresh
            self.last = self._now()
            return True
        return False


def invalidate(model):
    """Tell the inference engines of ``model`` that its weights were written behind autograd's
    back (native training steps)."""
    model._egn_generation = getattr(model, '_egn_generation', 0) + 1
    model._engine = None


# ---------------------------------------------------------------------------
# HRNet
# ---------------------------------------------------------------------------
def _act_of(seq):
    return ACT_RELU if any(isinstance(m, nn.ReLU) for m in seq) else ACT_NONE


class HRNetEngine(object):
    """Runs ``PoseHighResolutionNet`` (eval mode) on one GPU as HIP kernels."""

    def __init__(self, model):
        import os
        self.model = model
        self.programs = {}        # (device, N, H, W, decode) -> Program
        self._stamp = _Stamp(model)
        self.lanes = os.environ.get('EGONET_AMD_LANES', '1') != '0'   # branch-level concurrency
        # batches up to this size replay their program as one hipGraph (_forward_graphed); 0 = never.  OFF by default:
        # replays are bit-identical and faster (16 crops: 5.16 -> 5.01 ms per step; 64 crops: 12.79 -> 12.60), but inside
        # the GPU test suite hipGraphLaunch crashed the interpreter after the autograd-bridge tests had run in the same
        # process (profiles/r5_graph_replay_crash.txt) -- not root-caused, so nothing depends on it
        self.graph_max_n = int(os.environ.get('EGONET_AMD_GRAPH_MAX_N', '0'))
        self.fuse_layer1 = os.environ.get('EGONET_AMD_PW_FUSE', '1') != '0'      # layer1's 1x1 pairs on csrc/conv_pw.hip
        # fuse output i -> branch i of the next module on the same lane, no join in between
        self.chain_regions = os.environ.get('EGONET_AMD_CHAIN', '1') != '0'

    # -- recording ---------------------------------------------------------
    def _block(self, r, x, blk, tag):
        y = x
        for i in range(1, blk.depth + 1):
            conv = getattr(blk, 'conv%d' % i)
            bn = getattr(blk, 'bn%d' % i)
            k = conv.kernel_size[0]
            if i < blk.depth:
                y = r.conv(y, conv.weight, None, bn, ACT_RELU, None, conv.stride[0], conv.padding[0],
                           tag='%s.conv%d' % (tag, i))
            else:
                if blk.downsample is None:
                    res = x
                else:
                    pc, pb = blk.downsample[0], blk.downsample[1]
                    res = r.conv(x, pc.weight, None, pb, ACT_NONE, None, pc.stride[0], 0,
                                 tag=tag + '.downsample')
                y = r.conv(y, conv.weight, None, bn, ACT_RELU, res, conv.stride[0], conv.padding[0],
                           tag='%s.conv%d' % (tag, i))
        return y

    def _layer1(self, r, x, blocks):
        """layer1 (hrnet.py:325, 512-529: four Bottlenecks, 64 -> 256 channels).  Where the recorder has ``pw_pair`` (the
        inference recorder; the training tape walks the blocks layer by layer) each block's conv3 + residual + ReLU and
        the NEXT block's conv1 + ReLU are ONE launch of csrc/conv_pw.hip, the downsample conv and the last conv3 the same
        kernel's one-product form: the 256-channel tensor crosses HBM once per direction (EGONET_AMD_PW_FUSE=0: the
        general conv kernels, one launch per layer)."""
        def plain():
            t = x
            for k, blk in enumerate(blocks):
                t = self._block(r, t, blk, 'layer1.%d' % k)
            return t

        def is11(conv, cout, cin):
            return tuple(conv.weight.shape) == (cout, cin, 1, 1) and conv.stride[0] == 1 and conv.bias is None
        ok = self.fuse_layer1 and hasattr(r, 'pw_pair') and x.c == 64 and x.cs == 64 and (x.n * x.h * x.w) % 32 == 0 \
            and x.n * x.h * x.w * 256 * 4 < 2 ** 31 \
            and len(blocks) >= 1 and all(getattr(b, 'depth', 0) == 3 for b in blocks)
        if ok:
            for k, b in enumerate(blocks):
                ok = ok and is11(b.conv1, 64, 64 if k == 0 else 256) and is11(b.conv3, 256, 64) and \
                    tuple(b.conv2.weight.shape) == (64, 64, 3, 3) and b.conv2.stride[0] == 1
                if k == 0:
                    ok = ok and b.downsample is not None and is11(b.downsample[0], 256, 64)
                else:
                    ok = ok and b.downsample is None
        if not ok:
            return plain()
        b0 = blocks[0]
        h = r.conv(x, b0.conv1.weight, None, b0.bn1, ACT_RELU, None, 1, 0, tag='layer1.0.conv1')
        t = x
        for k, b in enumerate(blocks):
            h = r.conv(h, b.conv2.weight, None, b.bn2, ACT_RELU, None, 1, 1, tag='layer1.%d.conv2' % k)
            if k == 0:
                res, _ = r.pw_pair(t, b.downsample[0], b.downsample[1], None, False, tag='layer1.0.downsample')
            else:
                res = t
            nxt = blocks[k + 1] if k + 1 < len(blocks) else None
            t, h = r.pw_pair(h, b.conv3, b.bn3, res, True, nxt.conv1 if nxt is not None else None,
                             nxt.bn1 if nxt is not None else None, tag='layer1.%d.conv3' % k)
        return t

    def _unit_chain(self, r, x, seq_of_units, tag):
        """Sequential of Sequential(conv, bn[, relu]) (transition / fuse down paths)."""
        t = x
        for s, unit in enumerate(seq_of_units):
            conv, bn = unit[0], unit[1]
            t = r.conv(t, conv.weight, None, bn, _act_of(unit), None, conv.stride[0], conv.padding[0],
                       tag='%s.%d' % (tag, s))
        return t

    def _module(self, r, xs, mod, tag, region_open=False, keep_open=False):
        """``region_open``: the previous module left its fuse region open -- output i was
        produced on lane i, which is exactly what branch i of this module reads, so the
        branches continue on their lanes without a join/fork pair in between (lane 0 does
        not wait for the coarse lanes' fuse work).  ``keep_open``: leave this module's fuse
        region open for the next one (same stage, same branch count).  Returns (outs, open)."""
        xs = list(xs)
        # the resolution branches are independent (hrnet.py:286-287): one launch
        # lane each, so their kernels overlap each other's prologue / epilogue /
        # tail and the small coarse-branch grids do not leave the chip idle
        lanes = self.lanes and mod.num_branches > 1
        if lanes and not region_open:
            r.fork()
        for b, branch in enumerate(mod.branches):
            if lanes:
                r.lane(b)
            for k, blk in enumerate(branch):
                xs[b] = self._block(r, xs[b], blk, '%s.branches.%d.%d' % (tag, b, k))
        if lanes:
            r.join()
        if mod.fuse_layers is None:
            return xs, False
        outs = []
        if lanes:                       # the fuse outputs are independent too (hrnet.py:291-298)
            r.fork()
        for i, row in enumerate(mod.fuse_layers):
            if lanes:
                r.lane(i)
            terms = []
            for j in range(mod.num_branches):
                q = '%s.fuse_layers.%d.%d' % (tag, i, j)
                if row[j] is None:
                    terms.append((xs[j], 0))
                elif j > i:
                    conv, bn = row[j][0], row[j][1]
                    t = r.conv(xs[j], conv.weight, None, bn, ACT_NONE, None, 1, 0, tag=q)
                    terms.append((t, j - i))
                else:
                    terms.append((self._unit_chain(r, xs[j], row[j], q), 0))
            outs.append(r.fuse(terms, True, tag='%s.fuse%d' % (tag, i)))
        stay = lanes and keep_open and self.chain_regions and len(outs) == mod.num_branches
        if lanes and not stay:
            r.join()
        return outs, stay

    def _record(self, n, cin, h, w, decode_mode, r=None):
        """Walk the module tree once, issuing every layer to the recorder ``r``.
        The default recorder builds an inference program; egonet_amd.train_hrnet
        passes a tape that executes train-mode layers and records their backward."""
        m = self.model
        r = _Recorder() if r is None else r
        x = r.nchw_to_nhwc(Ref(SLOT_USER0, 0), n, cin, h, w, tag='input')
        t = r.conv(x, m.conv1.weight, None, m.bn1, ACT_RELU, None, 2, 1, tag='conv1')
        t = r.conv(t, m.conv2.weight, None, m.bn2, ACT_RELU, None, 2, 1, tag='conv2')
        t = self._layer1(r, t, m.layer1)
        ys = [t]
        for idx in (1, 2, 3):
            trans = getattr(m, 'transition%d' % idx)
            xs = []
            for i, tr in enumerate(trans):
                if tr is None:
                    xs.append(ys[i])
                elif isinstance(tr[0], nn.Conv2d):      # conv, bn, relu
                    xs.append(r.conv(ys[-1], tr[0].weight, None, tr[1], ACT_RELU, None, 1, 1,
                                     tag='transition%d.%d' % (idx, i)))
                else:
                    xs.append(self._unit_chain(r, ys[-1], tr, 'transition%d.%d' % (idx, i)))
            stage = getattr(m, 'stage%d' % (idx + 1))
            open_ = False
            for k, mod in enumerate(stage):
                nxt = stage[k + 1] if k + 1 < len(stage) else None
                keep = nxt is not None and nxt.num_branches == mod.num_branches
                xs, open_ = self._module(r, xs, mod, 'stage%d.%d' % (idx + 1, k), region_open=open_, keep_open=keep)
            ys = xs
        trunk = ys[0]
        J = m.num_joints
        mh, mw = trunk.h, trunk.w
        maps_ext = Ref(SLOT_USER0 + 1, 0)
        out_shapes = {'maps': (n, J, mh, mw)}
        if m.head_type == 'heatmap' and m.pixel_shuffle:
            # hrnet.py:373-383, 598-600: final_layer -> 1x1 conv + BN + ReLU -> PixelShuffle(up)
            fl, up = m.final_layer, m.upsample_layer
            t = r.conv(trunk, fl.weight, fl.bias, None, ACT_NONE, None, 1, fl.padding[0], tag='final_layer')
            f = int(m.upsamp_fact)
            u = r.conv(t, up[0].weight, up[0].bias, up[1], ACT_RELU, None, 1, 0, tag='upsample_layer.0')
            r.pixel_shuffle(u, J, f, maps_ext, tag='upsample_layer.3')
            mh, mw = mh * f, mw * f
            out_shapes['maps'] = (n, J, mh, mw)
            nslots = 2
        elif m.head_type == 'heatmap':
            fl = m.final_layer
            buf = Buf(n, mh, mw, J, cs=J, slot=SLOT_USER0 + 1, name='maps')
            r.conv(trunk, fl.weight, fl.bias, None, ACT_NONE, None, 1, fl.padding[0], dst=buf,
                   out_nchw=True, tag='final_layer')
            nslots = 2
        elif m.head_type == 'angleregression':
            # hrnet.py:384-422, 609-611: 1x1 conv (bias) -> 4 strided BasicBlocks -> AvgPool2d(4) ->
            # Linear + BatchNorm1d + ReLU -> Linear(256, 2).  The average pool and the first Linear
            # are ONE 4x4 valid convolution (filter = W[o][c] / 16 on every tap): the same sum.
            hd = m.head
            t = r.conv(trunk, hd[0].weight, hd[0].bias, None, ACT_NONE, None, 1, 0, tag='head.0')
            for k in range(1, 5):
                t = self._block(r, t, hd[k], 'head.%d' % k)
            pool = hd[5]
            ks = pool.kernel_size if isinstance(pool.kernel_size, int) else pool.kernel_size[0]
            if (t.h, t.w) != (ks, ks):
                raise ValueError('angle head expects a %dx%d map in front of AvgPool2d(%d), got %dx%d'
                                 % (ks, ks, ks, t.h, t.w))
            fc1, bn1, fc2 = m.final_fc[0], m.final_fc[1], m.final_fc[3]
            w1 = (fc1.weight.detach() / float(ks * ks)).view(fc1.out_features, fc1.in_features, 1, 1) \
                .expand(-1, -1, ks, ks).contiguous()
            y = r.conv(t, w1, fc1.bias, bn1, ACT_RELU, None, 1, 0, tag='final_fc.0')
            out = Buf(n, 1, 1, fc2.out_features, cs=fc2.out_features, slot=SLOT_USER0 + 1, name='angles')
            r.conv(y, fc2.weight.view(fc2.out_features, fc2.in_features, 1, 1), fc2.bias, None, ACT_NONE, None,
                   1, 0, dst=out, out_nchw=True, tag='final_fc.3')
            out_shapes = {'maps': (n, fc2.out_features)}          # the module returns [N, 2]
            nslots = 2
            decode_mode = None
        else:
            h1 = m.head1[0]
            aug = r.conv(trunk, h1.weight, h1.bias, None, ACT_NONE, None, 1, 0,
                         cout_cs=_round_up(J + 2, 4), tag='head1')
            r.ramps(aug, J, tag='coor_maps')
            r.nhwc_to_nchw(aug, J, maps_ext, tag='maps_nchw')
            aug.c = J + 2
            t = aug
            for k in range(4):
                t = self._block(r, t, m.head2[k], 'head2.%d' % k)
            fc = m.head2[4]
            if t.h != fc.kernel_size[0] or t.w != fc.kernel_size[1]:
                raise ValueError('coordinate head expects a %s map, got %dx%d'
                                 % (fc.kernel_size, t.h, t.w))
            cbuf = Buf(n, 1, 1, 2 * J, cs=2 * J, slot=SLOT_USER0 + 2, name='coords')
            r.conv(t, fc.weight, fc.bias, None, ACT_SIGMOID, None, 1, 0, dst=cbuf, out_nchw=True,
                   tag='head2.4')
            out_shapes['coords'] = (n, J, 2)
            nslots = 3
        if decode_mode is not None:
            r.decode(maps_ext, n, J, mh, mw, decode_mode, Ref(SLOT_USER0 + nslots, 0),
                     Ref(SLOT_USER0 + nslots + 1, 0), Ref(SLOT_USER0 + nslots + 2, 0), tag='decode')
            out_shapes['decode_slot'] = SLOT_USER0 + nslots
            nslots += 3
        return r, nslots, out_shapes

    def program(self, x, decode_mode=None, slot=0):
        """``slot``: independent copies of the program (own arena, own launch lanes) for batches IN FLIGHT together: a
        serving loop that alternates two streams runs slot 0 on one and slot 1 on the other, and the low-occupancy head
        and tail of one batch overlap the other batch's kernels."""
        if self._stamp.changed():
            self.programs.clear()
        n, c, h, w = x.shape
        key = (x.device, n, c, h, w, decode_mode) if not slot else (x.device, n, c, h, w, decode_mode, slot)
        prog = self.programs.get(key)
        if prog is None:
            if h % 32 or w % 32:
                raise ValueError('HRNet input height/width must be multiples of 32, got %dx%d' % (h, w))
            rec, nslots, shapes = self._record(n, c, h, w, decode_mode)
            prog = Program(rec, x.device, nslots)
            prog.out_shapes = shapes
            self.programs[key] = prog
        return prog

    # -- execution ---------------------------------------------------------
    def forward(self, x, decode_mode=None, timed=False, slot=0):
        """x [N,C,H,W] fp32 CUDA.  Returns what the module's forward returns;
        with decode_mode 0/1 additionally (xy[N,K,2], maxvals[N,K,1], idx[N,K])."""
        if x.dtype != torch.float32:
            raise TypeError('egonet_amd HRNet engine computes in fp32, got %s' % x.dtype)
        x = x.contiguous()
        nmax = self.max_batch(x.shape[1], x.shape[2], x.shape[3])
        if x.shape[0] > nmax:
            return self._forward_chunked(x, decode_mode, timed, slot, nmax)
        with torch.cuda.device(x.device):
            prog = self.program(x, decode_mode, slot)
            if not timed and x.shape[0] <= self.graph_max_n and not torch.cuda.is_current_stream_capturing():
                return self._forward_graphed(prog, x, decode_mode)
            prog.captured = False             # (the bindings below replace the static ones of a captured graph)
            shp = prog.out_shapes
            maps = torch.empty(shp['maps'], dtype=torch.float32, device=x.device)
            prog.bind(SLOT_USER0, x)
            prog.bind(SLOT_USER0 + 1, maps)
            outs = maps
            if 'coords' in shp:
                coords = torch.empty(shp['coords'], dtype=torch.float32, device=x.device)
                prog.bind(SLOT_USER0 + 2, coords)
                outs = (maps, coords)
            dec = None
            if decode_mode is not None:
                n, k = shp['maps'][:2]
                xy = torch.empty(n, k, 2, dtype=torch.float32, device=x.device)
                mx = torch.empty(n, k, 1, dtype=torch.float32, device=x.device)
                idx = torch.empty(n, k, dtype=torch.int32, device=x.device)
                s = shp['decode_slot']
                prog.bind(s, xy)
                prog.bind(s + 1, mx)
                prog.bind(s + 2, idx)
                dec = (xy, mx, idx)
            if timed:
                self.last_ms = prog.run_timed()
            else:
                prog.run()
        return outs if dec is None else (outs, dec)


    # the batch sizes the shipped tile table covers (tuned/gfx950.json): chunks of an oversized batch are cut to these
    CHUNK_SIZES = (128, 64, 32, 16, 8, 4, 2, 1)

    def max_batch(self, c, h, w):
        """Largest batch ONE program takes: the kernels address a tensor with 32-bit byte offsets (buffer instructions;
        csrc/conv_plan.hip refuses a tensor of 2 GiB), and the widest tensor of the network is layer1's 256-channel map at
        a quarter of the resolution (hrnet.py:512-529) or the stem's 64 channels at half of it: 4 MiB per 256 x 256 crop,
        i.e. 511 crops.  The reference takes any loader batch (libs/trainer/trainer.py:113-125): larger batches are cut
        into chunks (_forward_chunked), not refused.  EGONET_AMD_MAX_TENSOR_BYTES lowers the limit (tests)."""
        import os
        limit = int(os.environ.get('EGONET_AMD_MAX_TENSOR_BYTES', str(2 ** 31 - 1)))
        per_crop = 4 * max(256 * (h // 4) * (w // 4), 64 * (h // 2) * (w // 2), c * h * w)
        return max(1, limit // per_crop)

    def _forward_chunked(self, x, decode_mode, timed, slot, nmax):
        """A batch whose widest tensor would reach 2 GiB: cut into the largest table-covered chunk sizes that fit (at
        256 x 256: 128 crops), run chunk by chunk on the same stream, results concatenated -- bit-identical to calling
        the chunks one by one (tests/test_gpu_models.py::test_oversized_batches_are_chunked)."""
        n = x.shape[0]
        sizes, left = [], n
        for c in self.CHUNK_SIZES:
            if c > nmax:
                continue
            while left >= c:
                sizes.append(c)
                left -= c
        outs, decs, ms, at = [], [], 0.0, 0
        for c in sizes:
            r = self.forward(x[at:at + c], decode_mode, timed, slot)
            at += c
            if timed:
                ms += self.last_ms
            if decode_mode is not None:
                r, d = r
                decs.append(d)
            outs.append(r)
        if timed:
            self.last_ms = ms
        if isinstance(outs[0], tuple):
            out = tuple(torch.cat([o[k] for o in outs]) for k in range(len(outs[0])))
        else:
            out = torch.cat(outs)
        if decode_mode is None:
            return out
        return out, tuple(torch.cat([d[k] for d in decs]) for k in range(3))

    def _forward_graphed(self, prog, x, decode_mode):
        """Opt-in (EGONET_AMD_GRAPH_MAX_N=<n>, default 0).  Small batches (BASELINE configs[4]'s per-GPU shard, configs[0]'s
        single crop) are launch-bound -- ~320 launches of 5-15 us each: from its third run on, a program is replayed as
        ONE hipGraph (``egn_program_capture``: the launch lanes become graph edges).  Measured on MI355X
        (profiles/r5_small_batch.txt): 16 crops 5.68 -> 4.89 ms, 4 crops 4.22 -> 3.80, one crop 3.79 -> 3.62, outputs
        bit-identical.  A captured graph holds addresses, so the program owns static input / output tensors: the
        caller's crops are copied in (12.6 MB at 16 crops) and the outputs copied out -- the results are fresh tensors
        as in the eager path.  tools/inference.py:135-199 (the reference's per-image forward) is this regime."""
        shp = prog.out_shapes
        st = getattr(prog, 'static', None)
        if st is None:
            dev = x.device
            st = dict(x=torch.empty_like(x), maps=torch.empty(shp['maps'], dtype=torch.float32, device=dev))
            if 'coords' in shp:
                st['coords'] = torch.empty(shp['coords'], dtype=torch.float32, device=dev)
            if decode_mode is not None:
                n, k = shp['maps'][:2]
                st['xy'] = torch.empty(n, k, 2, dtype=torch.float32, device=dev)
                st['mx'] = torch.empty(n, k, 1, dtype=torch.float32, device=dev)
                st['idx'] = torch.empty(n, k, dtype=torch.int32, device=dev)
            prog.static, prog.runs, prog.captured = st, 0, False
        prog.bind(SLOT_USER0, st['x'])            # (same address: a captured graph stays valid)
        prog.bind(SLOT_USER0 + 1, st['maps'])
        if 'coords' in st:
            prog.bind(SLOT_USER0 + 2, st['coords'])
        if decode_mode is not None:
            s = shp['decode_slot']
            prog.bind(s, st['xy'])
            prog.bind(s + 1, st['mx'])
            prog.bind(s + 2, st['idx'])
        # (a caller that alternates streams on ONE program: the static input may still be read by the previous run)
        cur = torch.cuda.current_stream(x.device)
        prev = getattr(prog, 'last_stream', None)
        if prev is not None and prev != cur:
            cur.wait_stream(prev)
        prog.last_stream = cur
        st['x'].copy_(x)
        if prog.captured:
            prog.replay()
        else:
            prog.run()
            prog.runs += 1
            if prog.runs >= 2:                    # (short-lived programs -- nn.DataParallel replicas -- never pay a capture)
                # captured on a stream of its own: the caller's current stream may be the legacy default stream, which
                # cannot be captured; the graph is then launched on whatever stream is current
                cur = torch.cuda.current_stream(x.device)
                cs = torch.cuda.Stream(device=x.device)
                cs.wait_stream(cur)
                with torch.cuda.stream(cs):
                    prog.capture()
                cur.wait_stream(cs)
                prog.captured = True
        outs = st['maps'].clone()
        if 'coords' in st:
            outs = (outs, st['coords'].clone())
        if decode_mode is None:
            return outs
        return outs, (st['xy'].clone(), st['mx'].clone(), st['idx'].clone())


# ---------------------------------------------------------------------------
# lifter
# ---------------------------------------------------------------------------
class LifterEngine(object):
    """Runs ``FCModel`` (eval mode) as six fused GEMM launches (Linear + folded
    BatchNorm1d + ReLU (+ residual)); a Linear is a 1x1 convolution on
    [N,1,1,C]."""

    def __init__(self, model):
        self.model = model
        self.programs = {}
        self._stamp = _Stamp(model)

    def _record(self, n, ld_in=None):
        m = self.model
        act = ACT_LEAKY if m.leaky else ACT_RELU
        r = _Recorder()
        cin = m.w1.in_features
        if ld_in is None:
            x = r.nchw_to_nhwc(Ref(SLOT_USER0, 0), n, cin, 1, 1, tag='input')
        else:                      # caller provides [n, ld_in] rows already padded
            x = Buf(n, 1, 1, cin, cs=ld_in, slot=SLOT_USER0, name='input')

        def lin(t, fc, bn, a, res, tag, dst=None, nchw=False):
            return r.conv(t, fc.weight.view(fc.out_features, fc.in_features, 1, 1), fc.bias, bn, a,
                          res, 1, 0, dst=dst, out_nchw=nchw, tag=tag)

        y = lin(x, m.w1, m.batch_norm1, act, None, 'w1')
        for b, blk in enumerate(m.res_blocks):
            z = lin(y, blk.w1, blk.batch_norm1, act, None, 'res%d.w1' % b)
            y = lin(z, blk.w2, blk.batch_norm2, act | ACT_RES_AFTER, y, 'res%d.w2' % b)
        out = Buf(n, 1, 1, m.w2.out_features, cs=m.w2.out_features, slot=SLOT_USER0 + 1, name='out')
        lin(y, m.w2, None, ACT_NONE, None, 'w2', dst=out, nchw=True)
        return r

    def program(self, device, n, ld_in=None, slot=0):
        if self._stamp.changed():
            self.programs.clear()
        key = (device, n, ld_in) if not slot else (device, n, ld_in, slot)
        prog = self.programs.get(key)
        if prog is None:
            prog = Program(self._record(n, ld_in), device, 2)
            self.programs[key] = prog
        return prog

    def forward(self, x, ld_in=None, slot=0):
        """x [N,in] fp32 CUDA (or [N,ld_in] zero-padded rows) -> [N,out]."""
        if x.dtype != torch.float32:
            raise TypeError('egonet_amd lifter engine computes in fp32, got %s' % x.dtype)
        x = x.contiguous()
        n = x.shape[0]
        out = torch.empty(n, self.model.w2.out_features, dtype=torch.float32, device=x.device)
        if n == 0:
            return out
        with torch.cuda.device(x.device):
            prog = self.program(x.device, n, ld_in, slot)
            prog.bind(SLOT_USER0, x)
            prog.bind(SLOT_USER0 + 1, out)
            prog.run()
        return out
