#!/usr/bin/env python
"""Weight-gradient kernel on the HRNet training shapes: error against float64 (small batch slice is
not possible for a reduction over the batch, so the check runs at a reduced batch) and time per launch.

    python tools/wgrad_probe.py [--batch 32] [--shape cin,cout,h,w ...]
    EGN_WGRAD_WINO=0 python tools/wgrad_probe.py      # the direct kernels of conv_wgrad.hip
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonet_amd import _lib  # noqa: E402

SHAPES = [(48, 48, 64, 64), (96, 96, 32, 32), (192, 192, 16, 16), (384, 384, 8, 8), (48, 96, 64, 64)]


def run(L, n, cin, cout, h, w, iters, check):
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(n, h, w, cin, device='cuda', generator=g)
    dy = torch.randn(n, h, w, cout, device='cuda', generator=g)
    need = L.egn_conv2d_wgrad_ws_bytes(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1)
    ws = torch.empty(need // 4, device='cuda')
    dw = torch.empty(cout, cin, 3, 3, device='cuda')
    st = _lib.current_stream()

    def launch():
        _lib.check(L.egn_conv2d_wgrad_f32(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), n, h, w, cin, cin, cout, cout,
                                          3, 3, 1, 1, _lib.ptr(ws), need, st))
    launch()
    torch.cuda.synchronize()
    err = None
    if check:
        wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device='cuda', requires_grad=True)
        torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wt, None, 1, 1).backward(dy.permute(0, 3, 1, 2).double())
        err = float((dw.double() - wt.grad).abs().max() / wt.grad.abs().max())
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * n * h * w * cin * cout * 9
    print('%d,%d,%d,%d,%d: %.1f us  %.1f TFLOP/s (direct-algorithm flops, incl. the partial reduction)%s'
          % (n, cin, cout, h, w, us, fl / us * 1e-6, '' if err is None else '  max rel err vs fp64 %.2e' % err),
          flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--shape', action='append')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--no-check', action='store_true')
    a = ap.parse_args()
    L = _lib.lib()
    shapes = [tuple(int(v) for v in s.split(',')) for s in a.shape] if a.shape else SHAPES
    print('EGN_WGRAD_WINO=%s' % os.environ.get('EGN_WGRAD_WINO', '(unset: Winograd where it applies)'))
    for cin, cout, h, w in shapes:
        run(L, a.batch, cin, cout, h, w, a.iters, not a.no_check)


if __name__ == '__main__':
    main()
