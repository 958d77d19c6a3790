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


if __name__ == '__main__':
    args = sys.argv[1:]
    cfg = 78
    if args and args[0].startswith('--cfg='):
        cfg = int(args.pop(0)[6:])
    shapes = [tuple(int(v) for v in s.split(',')) for s in args] or [(64, 64, 64, 48, 48), (64, 32, 32, 96, 96)]
    torch.cuda.set_device(0)
    for s in shapes:
        run(s, cfg)
