#!/usr/bin/env python
"""Winograd (conv_wino.hip, cfg 45/46) vs direct conv kernels on one shape: max |diff| of the two
HIP paths on the same data and within-process A/B timing.

    python tools/wino_probe.py --shape 64,64,64,48,48 --direct 44 --iters 20
shape = N,H,W,Cin,Cout (3x3, stride 1, pad 1)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonet_amd import _lib, engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', action='append', required=True)
    ap.add_argument('--direct', default='0', help='direct cfg per shape (comma list) or one for all')
    ap.add_argument('--wino', default='45')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--res', type=int, default=1)
    ap.add_argument('--data', default='randn', choices=['randn', 'zeros', 'ones'],
                    help='operand VALUES (timing only): zeros / ones toggle far fewer datapath bits than random data -- '
                         'the difference is the power-management (clock) share of a kernel time')
    a = ap.parse_args()
    L = _lib.lib()
    torch.cuda.set_device(0)
    st = _lib.current_stream()
    dcfgs = [int(v) for v in a.direct.split(',')]
    for si, shp in enumerate(a.shape):
        n, h, w, cin, cout = [int(v) for v in shp.split(',')]
        g = torch.Generator().manual_seed(si)
        x = torch.randn(n, h, w, cin, generator=g).cuda()
        wt = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (3 * cin ** 0.5))
        wd = engine.pack_conv_weight(wt).cuda()
        wu = engine.pack_wino_weight(wt).cuda()
        wu43 = engine.pack_wino43_weight(wt).cuda() if cout % 48 == 0 and cin % 4 == 0 else None
        wu4 = engine.pack_wino4_weight(wt).cuda() if cout % 48 == 0 and cin % 8 == 0 else None
        if a.data != 'randn':
            fill = 0.0 if a.data == 'zeros' else 1.0
            x.fill_(fill); wd.fill_(fill); wu.fill_(fill)
            if wu43 is not None:
                wu43.fill_(fill)
            if wu4 is not None:
                wu4.fill_(fill)
        sc = (torch.rand(cout, generator=g) + 0.5).cuda()
        sh = torch.randn(cout, generator=g).cuda()
        res = torch.randn(n, h, w, cout, generator=g).cuda() if a.res else None
        if res is not None and a.data != 'randn':
            res.zero_()
        flops = 2.0 * n * h * w * cout * cin * 9
        dcfg = dcfgs[si] if len(dcfgs) > 1 else dcfgs[0]
        cases = [('direct%d' % dcfg, dcfg, wd)] + [('wino%s' % c, int(c), wu4 if int(c) >= 70 else {2: wu43, 3: wu4}.get(L.egn_conv_config_kind(int(c)), wu))
                                                    for c in a.wino.split(',')]
        outs, ok = {}, []
        for name, cfg, wp in cases:
            y = torch.full((n, h, w, cout), float('nan'), device='cuda')

            def launch(cfg=cfg, wp=wp, y=y):
                return L.egn_conv2d_f32(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(res),
                                        _lib.ptr(y), n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 1, 0, cfg, st)
            rc = launch()
            torch.cuda.synchronize()
            if rc:
                print('%s %s: rc %d' % (shp, name, rc), flush=True)
                continue
            outs[name] = y
            ok.append((name, launch))
        names = list(outs)
        if len(names) > 1:
            ref = outs[names[0]]
            # fp64 reference of a slice (image 0) for an absolute error figure
            xr = x[:1].permute(0, 3, 1, 2).double().cpu()
            yr = torch.nn.functional.conv2d(xr, wt.double(), padding=1) * sc.double().cpu().view(1, -1, 1, 1) \
                + sh.double().cpu().view(1, -1, 1, 1)
            if res is not None:
                yr = yr + res[:1].permute(0, 3, 1, 2).double().cpu()
            yr = torch.relu(yr).permute(0, 2, 3, 1)
            for nm in names:
                d = (outs[nm] - ref).abs().max().item()
                e64 = (outs[nm][:1].double().cpu() - yr).abs().max().item()
                print('%s %s: max|y - %s| = %.3e  max|y - fp64| (image 0) = %.3e  nan = %d' % (
                    shp, nm, names[0], d, e64, int(torch.isnan(outs[nm]).sum())), flush=True)
        best = {nm: None for nm, _ in ok}
        for _ in range(a.rounds):
            for nm, launch in ok:
                launch()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    launch()
                e1.record()
                e1.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / a.iters
                best[nm] = us if best[nm] is None else min(best[nm], us)
        for nm, _ in ok:
            print('%s %s: %.1f us  %.1f TFLOP/s (direct-algorithm flops)' % (shp, nm, best[nm], flops / best[nm] / 1e6),
                  flush=True)


if __name__ == '__main__':
    main()
