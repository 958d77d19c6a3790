#!/usr/bin/env python
"""Experiment: does running the backbone as K independent sub-batches on K
streams (desynchronised kernels fill each other's prologue / epilogue / tail
gaps) beat one launch list over the whole batch?

    python tools/split_probe.py --batch 64 --splits 1,2,4
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonet_amd import configs, synth, engine  # noqa: E402
from egonet_amd.model.heatmapModel import hrnet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--splits', default='1,2,4')
    ap.add_argument('--steps', type=int, default=10)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    cfg = configs.w48_config('heatmap')
    net = hrnet.get_pose_net(cfg, is_train=False)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=1))
    net = net.eval().cuda()
    x = synth.synth_crops(a.batch, 3, 256, 256, seed=1).cuda()
    for k in [int(v) for v in a.splits.split(',')]:
        engines = [engine.HRNetEngine(net) for _ in range(k)]
        streams = [torch.cuda.Stream() for _ in range(k)]
        parts = list(x.chunk(k))

        def step():
            for e, s, p in zip(engines, streams, parts):
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    e.forward(p, decode_mode=1)
            for s in streams:
                torch.cuda.current_stream().wait_stream(s)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print('splits %d: %.2f ms/step  %.0f crops/s' % (k, dt * 1e3, a.batch / dt), flush=True)


if __name__ == '__main__':
    main()
