#!/usr/bin/env python
"""Where a small batch's step goes (configs[4]'s 16-crop shard): for n crops, wall time of the eager
forward (one C call issuing ~320 launches on the launch lanes), the same program replayed as a hipGraph, and the sum of
the kernel times of one serial pass (hipEvents around every launch).  Shipped table, autotuning off.

    python tools/small_batch_probe.py [--batches 1,4,16] [--iters 100]
"""
import argparse
import os
import sys
import time

os.environ.setdefault('EGONET_AMD_AUTOTUNE', '0')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from egonet_amd import configs, synth  # noqa: E402
from egonet_amd.engine import SLOT_USER0  # noqa: E402
from egonet_amd.model.heatmapModel import hrnet  # noqa: E402


def wall(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', default='1,4,16')
    ap.add_argument('--iters', type=int, default=100)
    ap.add_argument('--head', default='heatmap')
    a = ap.parse_args()
    cfg = configs.w48_config(a.head)
    net = hrnet.get_pose_net(cfg, is_train=False)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=1))
    net = net.eval().cuda()
    eng = net._hip_engine()
    for n in [int(b) for b in a.batches.split(',')]:
        x = synth.synth_crops(n, 3, 256, 256, seed=7).cuda()
        eager = wall(lambda: eng.forward(x, decode_mode=1), a.iters)
        eng.forward(x, decode_mode=1, timed=True)
        ksum = float(eng.last_ms.sum())
        nl = int((eng.last_ms > 0).sum())
        # the same program with frozen bindings as one hipGraph
        prog = eng.program(x, 1)
        shp = prog.out_shapes
        keep = [torch.empty(shp['maps'], device='cuda')]
        prog.bind(SLOT_USER0, x)
        prog.bind(SLOT_USER0 + 1, keep[0])
        s = shp['decode_slot']
        nn_, k = shp['maps'][:2]
        for j, t in enumerate((torch.empty(nn_, k, 2, device='cuda'), torch.empty(nn_, k, 1, device='cuda'),
                               torch.empty(nn_, k, dtype=torch.int32, device='cuda'))):
            keep.append(t)
            prog.bind(s + j, t)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            prog.run()
            torch.cuda.synchronize()
            want = keep[0].clone()
            prog.capture()
            keep[0].zero_()
            graph = wall(prog.replay, a.iters)
            same = bool(torch.equal(keep[0], want))
        lanes0 = None
        print('n = %3d: eager %.3f ms  hipGraph replay %.3f ms (bit-identical: %s)  kernel sum %.3f ms over %d launches'
              % (n, eager, graph, same, ksum, nl), flush=True)
        del lanes0


if __name__ == '__main__':
    main()
