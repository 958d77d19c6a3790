#!/usr/bin/env python
"""Where do the device-to-device / host-to-device copies of one native HRNet training step come from?
torch.profiler with Python stacks around ONE step; prints the copy events grouped by the innermost frame
inside egonet_amd/.

    python tools/find_copies.py [--batch 32]
"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonet_amd import configs, synth                                  # noqa: E402
from egonet_amd.model.heatmapModel import hrnet                        # noqa: E402
from egonet_amd.train_hrnet import HRNetTrainStep                      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    a = ap.parse_args()
    os.environ.setdefault('EGONET_AMD_WGRAD_STREAM', '0')
    cfg = configs.w48_config('coordinates')
    net = hrnet.get_pose_net(cfg, is_train=False)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=1))
    net = net.cuda().train()
    tr = HRNetTrainStep(net, lr=1e-3)
    g = torch.Generator().manual_seed(100)
    x = synth.synth_crops(a.batch, 3, 256, 256, seed=50).cuda()
    tgt = torch.rand(a.batch, 33, 64, 64, generator=g).cuda()
    jt = (torch.rand(a.batch, 33, 2, generator=g) * 256).cuda()
    for _ in range(2):
        tr.step(x, tgt, jt)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        tr.step(x, tgt, jt)
        torch.cuda.synchronize()
    by = collections.Counter()
    names = collections.Counter()
    for ev in prof.events():
        nm = ev.name.lower()
        if 'memcpy' in nm or 'copy_' in nm or 'copybuffer' in nm or nm in ('aten::clone', 'aten::contiguous', 'aten::to'):
            names[ev.name] += 1
            frame = next((f for f in (ev.stack or []) if 'egonet_amd' in f), (ev.stack or ['?'])[0] if ev.stack else '?')
            by[(ev.name, frame)] += 1
    print('event names:', dict(names))
    for (nm, fr), n in by.most_common(25):
        print('%5d  %-28s %s' % (n, nm, fr))


if __name__ == '__main__':
    main()
