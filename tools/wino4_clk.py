#!/usr/bin/env python
"""Timeline of conv_wino4_kernel (cfg 78 = cfg 70 built with s_memtime stamps): cycles between the stamps of a
stage / an item, median over blocks and waves, per third of the waves (the thirds transform at different points).

    python tools/wino4_clk.py [--cfg=78|87] [N,H,W,Cin,Cout ...]      (87: conv_wino4w_kernel, 96 output channels per item)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonet_amd import _lib, engine  # noqa: E402

NTK, NW = 96, 12
STAGE = ['filter k-group 2s landed', 'issue loads / pieces (+ transform, third 0)', 'multiply g = 0',
         'filter k-group 2s+1 landed', 'issue loads (+ transform third 1), multiply g = 1 (+ transform third 2)',
         'own pieces landed, V written', 'barrier']


STAGE_W = ['filter k-group 4s landed', '(transform third 0 +) load k-group 4s+1', 'multiply k-group 0',
           'filter k-group 4s+1 landed', '(transform third 1 +) load, multiply k-group 1', 'wait, load, multiply k-group 2',
           'wait (+ transform third 2), load, halo pieces, multiply k-group 3', 'own pieces landed, V written', 'barrier']


def run(shape, cfg=78):
    L = _lib.lib()
    n, h, w, cin, cout = shape
    st = _lib.current_stream()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (3 * cin ** 0.5))
    wu = engine.pack_wino4_weight(wt).cuda()
    sc, sh = torch.ones(cout).cuda(), torch.zeros(cout).cuda()
    y = torch.empty(n, h, w, cout, device='cuda')
    nblk = 256
    stamps = torch.zeros(nblk * (NW * NTK + 1) * 2, dtype=torch.float32, device='cuda')
    for _ in range(3):
        rc = L.egn_conv2d_f32(_lib.ptr(x), _lib.ptr(wu), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(stamps), _lib.ptr(y),
                              n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 1, 0, cfg, st)
        assert rc == 0, rc
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        L.egn_conv2d_f32(_lib.ptr(x), _lib.ptr(wu), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(stamps), _lib.ptr(y),
                         n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 1, 0, cfg, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    t = stamps.cpu().numpy().view(np.uint64).reshape(nblk, NW * NTK + 1)
    nt = int(t[0, 0])
    wide = cfg == 87                 # conv_wino4w_kernel: 16-channel stages of four k-groups, 9 stamps per stage
    NS = 9 if wide else 7
    stage_names = STAGE_W if wide else STAGE
    mfma_issue = 6912 if wide else 3456
    S = cin // (16 if wide else 8)
    per_item = 5 + NS * S + 1 + 8
    tk = t[:, 1:].reshape(nblk, NW, NTK).astype(np.int64)
    print('shape %s: %d stamps per wave, %d stages per item, %d stamps per item; launch %.1f us' % (shape, nt, S, per_item, us))
    if nt < 1 + per_item:
        return
    b = 1            # item 0 starts at stamp 1 (stamp 0 = kernel start)
    med = lambda v: float(np.median(v))                     # noqa: E731
    names = ['item top -> own stage-0 pieces landed', '-> barrier', '-> stage 0 transformed', '-> barrier (K loop starts)']
    for k, nm in enumerate(names):
        print('  prologue  %-44s %7.0f' % (nm, med(tk[:, :, b + k + 1] - tk[:, :, b + k])))
    ks = b + 4       # stamp "K loop starts"
    for third in range(3):
        ws = slice(4 * third, 4 * third + 4)
        print('  waves %d..%d (transform third %d):' % (4 * third, 4 * third + 3, third))
        prev = tk[:, ws, ks]
        tot = np.zeros(NS)
        for s in range(S):
            for k in range(NS):
                cur = tk[:, ws, ks + 1 + NS * s + k]
                tot[k] += med(cur - prev)
                prev = cur
        for k in range(NS):
            print('    stage     %-72s %7.0f' % (stage_names[k], tot[k] / S))
        print('    stage total %.0f cycles   [MFMA issue of the SIMD\'s three waves: %d]' % (tot.sum() / S, mfma_issue))
    ke = ks + NS * S + 1      # "K loop done"
    print('  K loop: %.0f cycles' % med(tk[:, :, ke] - tk[:, :, ks]))
    names = ['round 0: accumulators written', 'exchange barrier', 'output transform, stores issued', 'barrier',
             'round 1: accumulators written', 'exchange barrier', 'output transform, stores issued', 'barrier']
    for k, nm in enumerate(names):
        print('  epilogue  %-44s %7.0f' % (nm, med(tk[:, :, ke + k + 1] - tk[:, :, ke + k])))
    print('  item: %.0f cycles' % med(tk[:, :, ke + 8] - tk[:, :, b]))


def run_half(shape, cfg=89):
    """conv_wino4h_kernel (cfg 89 = cfg 88 with stamps): 6 waves per block, two blocks per CU wanted.  Which blocks share
    a CU (XCC_ID, HW_ID: SE / SH / CU), whether the two were RESIDENT TOGETHER (the second's LDS base is not 0) or one
    after the other, the phase table of the blocks of either kind, and the span of a CU's four half-items."""
    L = _lib.lib()
    n, h, w, cin, cout = shape
    st = _lib.current_stream()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (3 * cin ** 0.5))
    wu = engine.pack_wino4_weight(wt).cuda()
    sc, sh = torch.ones(cout).cuda(), torch.zeros(cout).cuda()
    y = torch.empty(n, h, w, cout, device='cuda')
    dual = cfg == 91                   # conv_wino4d_kernel: one 12-wave workgroup per CU, its halves = the two blocks
    nblk, nw, ntk_ = (256, 12, 128) if dual else (512, 6, 128)
    stamps = torch.zeros(nblk * (nw * ntk_ + 2) * 2, dtype=torch.float32, device='cuda')

    def launch():
        return L.egn_conv2d_f32(_lib.ptr(x), _lib.ptr(wu), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(stamps), _lib.ptr(y),
                                n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 1, 0, cfg, st)
    for _ in range(3):
        assert launch() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    t = stamps.cpu().numpy().view(np.uint64).reshape(nblk, nw * ntk_ + 2)
    nt = int(t[0, 0])
    S = cin // 8
    per_item = 5 + 7 * S + 1 + 8
    nitems = min((nt - 2) // per_item, (ntk_ - 2) // per_item)
    tk = t[:, 2:].reshape(nblk, nw, ntk_).astype(np.int64)
    meta = t[:, 1]
    nstamps = t[:, 0]
    if dual:                           # a half = a block of the analysis below (same CU key for both)
        tk = tk.reshape(nblk * 2, 6, ntk_)
        meta = np.repeat(meta, 2)
        nstamps = np.repeat(nstamps, 2)
        nblk, nw = nblk * 2, 6
    ldsb, xcc, hwid = (meta >> 32).astype(np.int64), ((meta >> 16) & 0xf).astype(np.int64), (meta & 0xffff).astype(np.int64)
    print('shape %s skew %s: %d stamps per wave, %d stages, %d items per block analysed; launch %.1f us' %
          (shape, os.environ.get('EGN_W4H_SKEW', '0'), nt, S, nitems, us))
    used = nstamps > 0
    print('  LDS bases (HW_REG_LDS_ALLOC[11:0]) of the blocks that ran: %s' %
          dict(zip(*[v.tolist() for v in np.unique(ldsb[used], return_counts=True)])))
    med = lambda v: float(np.median(v)) if np.size(v) else float('nan')                     # noqa: E731
    b = 2
    key = (xcc << 16) | (hwid & 0xff00)                    # XCC | SE_ID[15:13] SH_ID[12] CU_ID[11:8]
    groups = {}
    for blk in np.nonzero(used)[0]:
        groups.setdefault(int(key[blk]), []).append(int(blk))
    print('  blocks per (XCC, SE, SH, CU) key: %s' %
          {i: int(c) for i, c in enumerate(np.bincount([len(v) for v in groups.values()])) if c})
    last = 2 + nitems * per_item - 1
    together, serial = [], []
    for kcu, blks in groups.items():
        if len(blks) != 2:
            continue
        a_, b_ = blks
        lo = max(tk[a_, 0, 0], tk[b_, 0, 0]); hi = min(tk[a_, 0, last], tk[b_, 0, last])
        (together if hi > lo else serial).append((kcu, blks))
    print('  CUs whose two blocks overlap in time: %d; one after the other: %d' % (len(together), len(serial)))

    def table(sel, title):
        if not len(sel):
            return
        sel = np.array(sel)
        print('  -- %s (%d blocks)' % (title, len(sel)))
        names = ['item top -> own stage-0 pieces landed', '-> barrier', '-> stage 0 transformed', '-> barrier (K loop starts)']
        for k, nm in enumerate(names):
            print('    prologue  %-44s %7.0f' % (nm, med(tk[sel][:, :, b + k + 1] - tk[sel][:, :, b + k])))
        ks = b + 4
        for third in range(3):
            ws = slice(2 * third, 2 * third + 2)
            prev = tk[sel][:, ws, ks]
            tot = np.zeros(7)
            for s_ in range(S):
                for k in range(7):
                    cur = tk[sel][:, ws, ks + 1 + 7 * s_ + k]
                    tot[k] += med(cur - prev)
                    prev = cur
            print('    waves %d..%d (third %d): %s  stage total %.0f' % (2 * third, 2 * third + 1, third,
                                                                     ' '.join('%5.0f' % (v / S) for v in tot), tot.sum() / S))
        ke = ks + 7 * S + 1
        print('    K loop: %.0f cycles' % med(tk[sel][:, :, ke] - tk[sel][:, :, ks]))
        names = ['round 0: accumulators written', 'exchange barrier', 'output transform, stores issued', 'barrier',
                 'round 1: accumulators written', 'exchange barrier', 'output transform, stores issued', 'barrier']
        for k, nm in enumerate(names):
            print('    epilogue  %-44s %7.0f' % (nm, med(tk[sel][:, :, ke + k + 1] - tk[sel][:, :, ke + k])))
        print('    item: %.0f cycles;  block (start -> end of its %d items): %.0f cycles' %
              (med(tk[sel][:, :, ke + 8] - tk[sel][:, :, b]), nitems, med(tk[sel][:, 0, last] - tk[sel][:, 0, 0])))
    table([bk for _, blks in together for bk in blks], 'blocks resident TOGETHER with their CU partner')
    table([bk for _, blks in serial for bk in blks], 'blocks that ran ALONE on their CU (partner before / after)')
    for title, grp in (('together', together), ('one after the other', serial)):
        if grp:
            span = [max(tk[bk, 0, last] for bk in blks) - min(tk[bk, 0, 0] for bk in blks) for _, blks in grp]
            print('  CU span of its %d half-items, blocks %s: %.0f cycles (median)' % (2 * nitems, title, med(span)))
    shown = 0
    for kcu, blks in sorted(together):
        if shown >= 3:
            break
        shown += 1
        t0 = min(int(tk[bk, 0, 0]) for bk in blks)
        for bk in blks:
            ev = []
            for it in range(nitems):
                o = 2 + it * per_item
                ev.append('item %d: K %6d..%6d end %6d' % (it, tk[bk, 0, o + 4] - t0, tk[bk, 0, o + 4 + 7 * S + 1] - t0,
                                                          tk[bk, 0, o + per_item - 1] - t0))
            print('    cu %06x block %3d ldsbase %4d: start %6d | %s' % (kcu, bk, ldsb[bk], tk[bk, 0, 1] - t0, ' | '.join(ev)))


if __name__ == '__main__':
    args = sys.argv[1:]
    cfg = 78
    if args and args[0].startswith('--cfg='):
        cfg = int(args.pop(0)[6:])
    shapes = [tuple(int(v) for v in s.split(',')) for s in args] or [(64, 64, 64, 48, 48), (64, 32, 32, 96, 96)]
    torch.cuda.set_device(0)
    for s in shapes:
        run_half(s, cfg) if cfg in (89, 91) else run(s, cfg)
