#!/usr/bin/env python
"""Launch one convolution shape with given tile configs repeatedly (for
rocprofv3 --pmc / --kernel-trace runs and quick within-process A/B timing).

    python tools/conv_probe.py --shape 64,64,64,48,48,3,1,1 --cfg 1,6 --iters 20
shape = N,H,W,Cin,Cout,k,stride,pad
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonet_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', action='append', required=True)
    ap.add_argument('--cfg', default='0')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--rounds', type=int, default=1)
    ap.add_argument('--res', type=int, default=1)
    a = ap.parse_args()
    L = _lib.lib()
    torch.cuda.set_device(0)
    st = _lib.current_stream()
    for shp in a.shape:
        n, h, w, cin, cout, k, s, p = [int(v) for v in shp.split(',')]
        cs_in, cs_out = (cin + 3) // 4 * 4, (cout + 3) // 4 * 4
        coutp, nchunk = (cout + 15) // 16 * 16, (cin + 15) // 16
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        x = torch.randn(n * h * w * cs_in, device='cuda')
        wt = torch.randn(nchunk * k * k * 4 * coutp * 4, device='cuda') * 0.05
        sc, sh = torch.ones(coutp, device='cuda'), torch.zeros(coutp, device='cuda')
        y = torch.empty(n * ho * wo * cs_out, device='cuda')
        res = torch.randn(n * ho * wo * cs_out, device='cuda') if a.res else None
        flops = 2.0 * n * ho * wo * cout * cin * k * k
        cfgs = [int(c) for c in a.cfg.split(',')]

        def launch(cfg):
            return L.egn_conv2d_f32(_lib.ptr(x), _lib.ptr(wt), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(res),
                                    _lib.ptr(y), n, h, w, cin, cs_in, cout, cs_out, k, k, s, p, 1, 0, cfg, st)
        ok = []
        for cfg in cfgs:
            rc = launch(cfg)
            if rc:
                print('%s cfg %d: rc %d' % (shp, cfg, rc))
            else:
                launch(cfg)
                ok.append(cfg)
        torch.cuda.synchronize()
        best = {c: None for c in ok}
        for _ in range(a.rounds):              # interleaved rounds: within-process A/B
            for cfg in ok:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    launch(cfg)
                e1.record()
                e1.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / a.iters
                best[cfg] = us if best[cfg] is None else min(best[cfg], us)
        for cfg in ok:
            print('%s cfg %d: %.1f us  %.1f TFLOP/s' % (shp, cfg, best[cfg], flops / best[cfg] / 1e6), flush=True)


if __name__ == '__main__':
    main()
