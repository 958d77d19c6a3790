#!/usr/bin/env python
"""Timeline of conv_wino8_kernel (cfg 58 = cfg 51 built with s_memtime stamps): where do the cycles of a
K step / an item go -- waiting for the own DMA, waiting at the barrier, transform + MFMA loop, exchange, epilogue.

    python tools/wino_clk.py [N,H,W,Cin,Cout ...]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonet_amd import _lib, engine  # noqa: E402

NTK = 48


def run(shape, cfg, nw=8, kch=16, per_cu=1):
    """nw waves per block, kch channels per K stage, per_cu blocks per CU (cfg 69: 4 waves, 8 channels, 2 blocks)."""
    L = _lib.lib()
    n, h, w, cin, cout = shape
    st = _lib.current_stream()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (3 * cin ** 0.5))
    wu = engine.pack_wino_weight(wt).cuda()
    sc = torch.ones(cout).cuda()
    sh = torch.zeros(cout).cuda()
    y = torch.empty(n, h, w, cout, device='cuda')
    nblk = 256 * per_cu
    stamps = torch.zeros(nblk * (nw * NTK + 1) * 2, dtype=torch.float32, device='cuda')     # u64 view
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
    t = stamps.cpu().numpy().view(np.uint64).reshape(nblk, nw * NTK + 1)
    nt = int(t[0, 0])
    nchunk = cin // kch
    per_item = 3 * nchunk + 3
    print('cfg %d shape %s: %d stamps per wave, %d K steps per item, %d items in the stamp window' % (
        cfg, shape, nt, nchunk, (min(nt, NTK) - 1) // per_item))
    tk = t[:, 1:].reshape(nblk, nw, NTK).astype(np.int64)
    used = min(nt, NTK)
    nitems = (used - 1) // per_item
    if nitems == 0:
        return
    # phases per K step: [top -> own DMA landed], [-> past barrier], [-> next top / K loop done]
    ph = {'wait own DMA': [], 'wait barrier': [], 'transform + MFMA': [], 'out transform + exchange': [],
          'epilogue (res + stores issued)': [], 'item head (-> first step top)': []}
    for it in range(nitems):
        base = 1 + it * per_item
        for c in range(nchunk):
            a = base + 3 * c
            ph['wait own DMA'].append(tk[:, :, a + 1] - tk[:, :, a])
            ph['wait barrier'].append(tk[:, :, a + 2] - tk[:, :, a + 1])
            ph['transform + MFMA'].append(tk[:, :, a + 3] - tk[:, :, a + 2])
        k = base + 3 * nchunk
        ph['out transform + exchange'].append(tk[:, :, k + 1] - tk[:, :, k])
        ph['epilogue (res + stores issued)'].append(tk[:, :, k + 2] - tk[:, :, k + 1])
        ph['item head (-> first step top)'].append(tk[:, :, base] - tk[:, :, base - 1])
    item_span = tk[:, :, 1 + nitems * per_item - 1 + 0] - tk[:, :, 0]
    tot = float(np.median(item_span)) / nitems
    print('  cycles per item (median over blocks x waves): %.0f   [MFMA issue time of one wave: %d, of the SIMD\'s two: %d]'
          % (tot, nchunk * 6 * kch * 32, nchunk * 6 * kch * 64))
    for name, v in ph.items():
        a = np.stack(v)            # [samples, blocks, waves]
        per = a.sum(axis=0) / nitems
        print('  %-34s per item: median %7.0f  p10 %7.0f  p90 %7.0f   (%4.1f %% of the item)  per occurrence %6.0f' % (
            name, np.median(per), np.percentile(per, 10), np.percentile(per, 90), 100 * np.median(per) / tot,
            np.median(a)))
    # the two waves of a SIMD (w, w+4): how far apart are they at the step tops?
    a0 = 1
    if nw == 8:
        skew = tk[:, 4:, a0 + 2] - tk[:, :4, a0 + 2]
        print('  barrier exit skew wave w+4 vs w (first step): median %d  |max| %d' % (np.median(skew), np.abs(skew).max()))
    # the stamped window of a block in cycles, the whole launch in microseconds (hipEvents, 10 launches): the launch
    # runs nitems_total / nitems_window of the window per block -> the clock the shader counter ran at
    print('  stamped window (%d items): median %d cycles per block; launch %.1f us' % (
        nitems, np.median(item_span), us))


if __name__ == '__main__':
    shapes = [tuple(int(v) for v in s.split(',')) for s in sys.argv[1:]] or \
        [(64, 64, 64, 48, 48), (64, 32, 32, 96, 96), (64, 16, 16, 192, 192)]
    torch.cuda.set_device(0)
    for s in shapes:
        for cfg in (58, 63):          # conv_wino8_kernel / conv_wino9_kernel stamp builds
            run(s, cfg)
        run(s, 69, nw=4, kch=8, per_cu=2)     # conv_wino9_kernel, 8-channel stages, two 4-wave blocks per CU
