#!/usr/bin/env python
"""Measure tile configurations for the batch sizes BASELINE.json names besides the bench's own:
configs[0] (1 crop), the 4-crop reference fixtures, configs[4]'s per-GPU shard (16 crops) and its single-GPU form
(128 crops) -- so that `tuned/gfx950.json` decides every conv shape of those programs and nothing is autotuned on the
box at first use (a multi-second stall, a selection that is not reproducible) or left to the cost model.

    python tools/tune_sizes.py --sizes 1,2,4,8,16,128 --out gpurun_out/gfx950.json

Builds the HRNet-W48 programs (both heads) and the lifter program for each size with autotuning ON; the tuner times
every configuration that plans for a shape (tuner.tune) and the merged table is written to --out.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from egonet_amd import configs, synth, tuner                         # noqa: E402
from egonet_amd.model.heatmapModel import hrnet as hip_hrnet          # noqa: E402
from egonet_amd.model import FCmodel as hip_fc                        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sizes', default='1,2,4,8,16,128')
    ap.add_argument('--out', required=True)
    ap.add_argument('--heads', default='coordinates,heatmap')
    a = ap.parse_args()
    os.environ['EGONET_AMD_AUTOTUNE'] = '1'
    os.environ.pop('EGONET_AMD_WINO', None)
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    sizes = [int(v) for v in a.sizes.split(',')]
    for head in a.heads.split(','):
        cfg = configs.w48_config(head)
        net = hip_hrnet.get_pose_net(cfg, is_train=False)
        net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=1))
        net = net.eval().cuda()
        for n in sizes:
            t0 = time.time()
            x = synth.synth_crops(n, 3, 256, 256, seed=3).cuda()
            with torch.no_grad():
                net(x)
                net._hip_engine().forward(x, decode_mode=1)
            torch.cuda.synchronize()
            net._engine = None                     # drop the programs (arena) of this size
            print('%s n=%d: %d shapes tuned so far, %.1f s' % (head, n, len(tuner.tuned_in_process()), time.time() - t0),
                  flush=True)
    lif = hip_fc.get_fc_model(1, configs.w48_config(), 66, 96)
    lif.load_state_dict(synth.synth_state_dict(lif.state_dict(), seed=2))
    lif = lif.eval().cuda()
    for n in sizes:
        with torch.no_grad():
            lif(torch.randn(n, 66, device='cuda'))
            # the pipeline hands the lifter rows padded to 68 floats (model/egonet.py: ld_in)
            lif._hip_engine().forward(torch.zeros(n, 68, device='cuda'), ld_in=68)
    torch.cuda.synchronize()
    merged = dict(tuner._load())
    merged.update(tuner.tuned_in_process())
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, 'w') as f:
        json.dump(merged, f, indent=0, sort_keys=True)
    print('%d entries (%d new) -> %s' % (len(merged), len(tuner.tuned_in_process()), a.out))


if __name__ == '__main__':
    main()
