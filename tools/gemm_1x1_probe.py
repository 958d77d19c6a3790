#!/usr/bin/env python
"""Would layer1's 1x1 convolutions (hrnet.py:95-133: 64 -> 256 and 256 -> 64 on 64 x 64 maps) run faster as dense GEMMs
on csrc/gemm.hip (NT form: C[pixels][Cout] = X[pixels][Cin] . W[Cout][Cin]^T, the weight as it lies) than on the conv
kernels?  Times both on the bench's shapes (64 crops).  The GEMM launch has a bias-only epilogue (no BatchNorm scale,
residual or ReLU): a lower bound on what a fused-epilogue route could reach.

    python tools/gemm_1x1_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonet_amd import _lib, engine  # noqa: E402
from tools.gemm_probe import time_us  # noqa: E402


def main():
    L = _lib.lib()
    torch.cuda.set_device(0)
    st = _lib.current_stream()
    n, h, w = 64, 64, 64
    M = n * h * w
    for cin, cout, res in ((64, 256, True), (64, 256, False), (256, 64, False), (64, 64, False)):
        g = torch.Generator().manual_seed(cin)
        x = torch.randn(M, cin, generator=g).cuda()
        wt = (torch.randn(cout, cin, generator=g) / cin ** 0.5)
        wd = wt.cuda()
        bias = torch.randn(cout, generator=g).cuda()
        y = torch.empty(M, cout, device='cuda')
        r = torch.randn(M, cout, generator=g).cuda() if res else None
        wp = engine.pack_conv_weight(wt.view(cout, cin, 1, 1)).cuda()
        sc = torch.ones(max(cout, 16), device='cuda')
        gb = 4.0 * M * (cin + cout * (2 if res else 1)) / 1e9
        for cfg in (13, 18, 23, 28, 16, 17):
            def conv(cfg=cfg):
                return L.egn_conv2d_f32(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(sc), _lib.ptr(bias), _lib.ptr(r), _lib.ptr(y), n, h, w,
                                        cin, cin, cout, cout, 1, 1, 1, 0, 1, 0, cfg, st)
            if conv() != 0:
                continue
            us = time_us(conv, iters=10)
            print('%3d -> %3d res %d  conv cfg %2d: %7.1f us  %5.2f TB/s' % (cin, cout, int(res), cfg, us, gb / us * 1e3), flush=True)
        for v in (0, 1, 2, 3):
            if not L.egn_gemm_supported(0, M, cout, cin, cin, cin, cout):
                continue
            if v != 2 and cout % 128:
                continue

            def gemm(v=v):
                return L.egn_gemm_f32(0, _lib.ptr(x), _lib.ptr(wd), _lib.ptr(y), _lib.ptr(bias), M, cout, cin, cin, cin, cout, v,
                                      None, 0, st)
            if gemm() != 0:
                continue
            gb0 = 4.0 * M * (cin + cout) / 1e9
            us = time_us(gemm, iters=10)
            print('%3d -> %3d        gemm NT v%d: %7.1f us  %5.2f TB/s (bias-only epilogue, no residual)' % (cin, cout, v, us, gb0 / us * 1e3),
                  flush=True)


if __name__ == '__main__':
    main()
