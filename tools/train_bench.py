"""BASELINE config 3: FC lifter forward + backward + Adam, batch 4096 2D->3D
key-point sets on one MI355X, native HIP path (egonet_amd.train_lifter).

    python tools/train_bench.py [--batch 4096] [--steps 50] [--warmup 5] [--graph]

Prints one JSON line: sets/s, ms/step and the GEMM flop rate (3 GEMMs per
Linear: forward, dgrad, wgrad; 2*M*N*K each).  The CPU oracle's rate on the same
batch is printed by tests/cpu_baselines.py (test infrastructure).
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from egonet_amd import configs, synth                     # noqa: E402
from egonet_amd.model import FCmodel                       # noqa: E402
from egonet_amd.train_lifter import LifterTrainStep        # noqa: E402


def gemm_flops(net, batch):
    total = 0
    lin = [m for m in net.modules() if isinstance(m, torch.nn.Linear)]
    for i, m in enumerate(lin):
        per = 2.0 * batch * m.in_features * m.out_features
        total += per * (2 if i == 0 else 3)        # no dgrad into the network input
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4096)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--dropout', type=float, default=0.5)
    ap.add_argument('--graph', action='store_true', help='replay the step as one hipGraph')
    a = ap.parse_args()
    cfg = configs.w48_config()
    cfg['FCModel']['dropout'] = a.dropout
    net = FCmodel.get_fc_model(1, cfg, 66, 96)
    sd = synth.synth_state_dict(net.state_dict(), seed=2)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(a.batch, 66, generator=g), torch.randn(a.batch, 96, generator=g)
    net = net.cuda().train()
    tr = LifterTrainStep(net, lr=1e-3)
    xd, yd = x.cuda(), y.cuda()
    for _ in range(a.warmup):
        tr.step(xd, yd)
    torch.cuda.synchronize()
    step = lambda: tr.step(xd, yd)                      # noqa: E731
    if a.graph:
        # the whole iteration (~130 launches, static shapes) as ONE hipGraph launch
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            tr.step(xd, yd)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            captured_loss = tr.step(xd, yd)
        step = lambda: (graph.replay(), captured_loss)[1]      # noqa: E731
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        loss = step()
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.steps
    dev_ms = e0.elapsed_time(e1) / a.steps
    fl = gemm_flops(net, a.batch)
    out = {
        'metric': 'lifter_train_sets_per_sec', 'value': round(a.batch / wall, 1), 'unit': 'sets/s',
        'ms_per_step': round(wall * 1e3, 4), 'device_ms_per_step': round(dev_ms, 4),
        'gemm_tflops': round(fl / (dev_ms * 1e-3) / 1e12, 2), 'gemm_gflop_per_step': round(fl / 1e9, 2),
        'dtype': 'f32', 'loss': float(loss.item()),
        'config': {'workload': 'train_lifting FCModel(66->96, 1024x2 blocks) fwd+bwd+Adam', 'batch': a.batch,
                   'dropout': a.dropout, 'steps': a.steps, 'warmup': a.warmup, 'hipgraph': bool(a.graph)},
    }
    print(json.dumps(out))


if __name__ == '__main__':
    main()
