"""CPU rates of the training oracles on the workloads tools/train_bench.py and
tools/train_hc_bench.py time on the GPU (test infrastructure: the oracle is the
checker, here only timed so the GPU numbers have a host-side figure beside them).

    python tests/cpu_baselines.py [--lifter-batch 4096] [--hc-batch 2] [--threads 16]

Prints one JSON line per workload.  Bounded samples: 3 lifter iterations, 1
HRNet-W48 iteration (forward + backward + Adam through torch autograd, fp32).
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from egonet_amd import configs, synth                                  # noqa: E402
from egonet_amd.model import FCmodel                                    # noqa: E402
from egonet_amd.model.heatmapModel import hrnet                         # noqa: E402
from oracle.hrnet_train_oracle import HRNetTrainOracle                  # noqa: E402
from oracle.lifter_train_oracle import LifterTrainOracle                # noqa: E402


def lifter(batch, n=3):
    cfg = configs.w48_config()
    net = FCmodel.get_fc_model(1, cfg, 66, 96)
    sd = synth.synth_state_dict(net.state_dict(), seed=2)
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(batch, 66, generator=g), torch.randn(batch, 96, generator=g)
    orc = LifterTrainOracle(sd, lr=1e-3)
    orc.step(x, y)
    t0 = time.perf_counter()
    for _ in range(n):
        orc.step(x, y)
    dt = (time.perf_counter() - t0) / n
    return {'metric': 'lifter_train_sets_per_sec', 'value': round(batch / dt, 1), 'unit': 'sets/s',
            'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d iterations of one %d-set batch (dropout off)' % (n, batch)}


def hc(batch, n=1):
    cfg = configs.w48_config('coordinates')
    net = hrnet.get_pose_net(cfg, is_train=False)
    sd = synth.synth_state_dict(net.state_dict(), seed=1)
    g = torch.Generator().manual_seed(100)
    x = synth.synth_crops(batch, 3, 256, 256, seed=50)
    tgt = torch.rand(batch, 33, 64, 64, generator=g)
    jt = torch.rand(batch, 33, 2, generator=g) * 256
    orc = HRNetTrainOracle(sd, cfg, lr=1e-3)
    t0 = time.perf_counter()
    for _ in range(n):
        orc.step(x, tgt, jt)
    dt = (time.perf_counter() - t0) / n
    return {'metric': 'hc_train_crops_per_sec', 'value': round(batch / dt, 3), 'unit': 'crops/s',
            'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d iteration(s) of one %d-crop batch, HRNet-W48 256x256' % (n, batch)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lifter-batch', type=int, default=4096)
    ap.add_argument('--hc-batch', type=int, default=2)
    ap.add_argument('--threads', type=int, default=min(16, os.cpu_count() or 1))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    print(json.dumps(lifter(a.lifter_batch)))
    if a.hc_batch > 0:
        print(json.dumps(hc(a.hc_batch)))


if __name__ == '__main__':
    main()
