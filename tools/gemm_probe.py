#!/usr/bin/env python
"""csrc/gemm.hip on the lifter's shapes: every form x tile variant against a float64 product (rocBLAS dgemm
through torch -- the checker, not the product) and its time per launch.

    python tools/gemm_probe.py [--B 4096] [--H 1024]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonet_amd import _lib  # noqa: E402


def time_us(fn, iters=30, rounds=3):
    best = None
    for _ in range(rounds):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        best = us if best is None else min(best, us)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=4096)
    ap.add_argument('--H', type=int, default=1024)
    ap.add_argument('--data', default='randn', choices=['randn', 'zeros'],
                    help='operand values: zeros toggle no datapath bits -- the time difference is the clock (power) share')
    a = ap.parse_args()
    L = _lib.lib()
    torch.cuda.set_device(0)
    st = _lib.current_stream()
    B, H = a.B, a.H
    g = torch.Generator().manual_seed(0)
    act = torch.randn(B, H, generator=g).cuda()          # activations / dz  [B][H]
    act2 = torch.randn(B, H, generator=g).cuda()
    W = (torch.randn(H, H, generator=g) / H ** 0.5).cuda()   # [out][in]
    bias = torch.randn(H, generator=g).cuda()
    if a.data == 'zeros':
        act.zero_(); act2.zero_(); W.zero_(); bias.zero_()
    cases = [
        ('NT fwd  z = a W^T + b', 0, act, W, bias, B, H, H, (act.double() @ W.double().t() + bias.double()), (0, 1, 2, 3)),
        ('NN dgrad da = dz W', 1, act, W, None, B, H, H, (act.double() @ W.double()), (0, 1, 2)),
        ('TN wgrad dW = dz^T a', 2, act, act2, None, H, H, B, (act.double().t() @ act2.double()), (0, 1)),
    ]
    for name, form, A, Bm, bs, M, N, K, want, variants in cases:
        flops = 2.0 * M * N * K
        need = L.egn_gemm_ws_bytes(form, M, N, K)
        ws = torch.empty(max(need // 4, 4), device='cuda')
        for v in variants:
            C = torch.full((M, N), float('nan'), device='cuda')

            def launch(v=v, C=C):
                return L.egn_gemm_f32(form, _lib.ptr(A), _lib.ptr(Bm), _lib.ptr(C), _lib.ptr(bs), M, N, K, A.shape[1],
                                      Bm.shape[1], N, v, _lib.ptr(ws), need, st)
            rc = launch()
            torch.cuda.synchronize()
            if rc:
                print('%-26s variant %d: rc %d' % (name, v, rc), flush=True)
                continue
            err = float((C.double() - want).abs().max())
            scale = float(want.abs().max())
            us = time_us(launch)
            print('%-26s variant %d: max|err| %.2e (max|C| %.1f, nan %d)   %7.1f us  %6.1f TFLOP/s' % (
                name, v, err, scale, int(torch.isnan(C).sum()), us, flops / us / 1e6), flush=True)
    # for scale: the library GEMM (hipBLASLt / rocBLAS through torch) on the same shapes
    for name, fn, fl in [('torch NT', lambda: torch.addmm(bias, act, W.t()), 2.0 * B * H * H),
                         ('torch NN', lambda: act @ W, 2.0 * B * H * H),
                         ('torch TN', lambda: act.t() @ act2, 2.0 * B * H * H)]:
        us = time_us(fn)
        print('%-26s            %7.1f us  %6.1f TFLOP/s   (library, for scale)' % (name, us, fl / us / 1e6), flush=True)


if __name__ == '__main__':
    main()
