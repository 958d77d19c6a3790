#!/usr/bin/env python
"""csrc/conv_pw.hip alone: layer1's 1x1 pair (and its one-product forms) at a given number of crops -- time by hipEvents,
algorithmic TB/s and TFLOP/s.  (rocprofv3 --pmc over this script: tools/pmc_pw.sh.)

    python tools/pw_probe.py [--crops 64] [--iters 20]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from egonet_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--crops', type=int, default=64)
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    L = _lib.lib()
    m = a.crops * 64 * 64
    g = torch.Generator().manual_seed(0)
    h = torch.randn(m, 64, generator=g).cuda()
    res = torch.randn(m, 256, generator=g).cuda()
    w3 = (torch.randn(256 * 64, generator=g) / 8).cuda()
    w1 = (torch.randn(64 * 256, generator=g) / 16).cuda()
    s3, s1 = torch.randn(256, generator=g).cuda(), torch.randn(64, generator=g).cuda()
    out, hn = torch.empty(m, 256, device='cuda'), torch.empty(m, 64, device='cuda')
    st = _lib.current_stream()
    cases = [('pair  (conv3 + residual + ReLU + conv1 + ReLU)', True, True, 1),
             ('single (conv3 + residual + ReLU)', False, True, 1),
             ('single (downsample: no residual, no ReLU)', False, False, 0)]
    for name, fused, use_res, relu1 in cases:
        def run():
            _lib.check(L.egn_pw_pair_f32(_lib.ptr(h), _lib.ptr(res) if use_res else None, _lib.ptr(w3), _lib.ptr(s3),
                                         _lib.ptr(w1) if fused else None, _lib.ptr(s1) if fused else None, _lib.ptr(out),
                                         _lib.ptr(hn) if fused else None, m, relu1, st))
        for _ in range(3):
            run()
        best = None
        for _ in range(a.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1)
            best = t if best is None or t < best else best
        nbytes = 4.0 * m * (64 + 256 * (2 if use_res else 1) + (64 if fused else 0))
        flops = 2.0 * m * 64 * 256 * (2 if fused else 1)
        print('%d crops  %-52s %7.1f us  %5.2f TB/s  %6.1f TFLOP/s' % (a.crops, name, best * 1e3, nbytes / best / 1e9,
                                                                       flops / best / 1e9), flush=True)


if __name__ == '__main__':
    main()
