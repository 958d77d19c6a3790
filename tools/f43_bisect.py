#!/usr/bin/env python
"""Which layers' F(4x4,3x3) kernels move the network output: the HRNet-W48 forward at the bench batch with
F(4x4,3x3) off, on, and on for one shape class at a time (EGONET_AMD_F43_MATCH), against the float64-free
F(2x2,3x3)-only run of the same engine.

    python tools/f43_bisect.py [--batch 64]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonet_amd import configs, synth                                  # noqa: E402
from egonet_amd.model.heatmapModel import hrnet                        # noqa: E402


def run(x, env):
    for k in ('EGONET_AMD_F43', 'EGONET_AMD_F43_MATCH'):
        os.environ.pop(k, None)
    os.environ.update(env)
    cfg = configs.w48_config('heatmap')
    net = hrnet.get_pose_net(cfg, is_train=False)
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), seed=1))
    net = net.eval().cuda()
    eng = net._hip_engine()
    maps, _ = eng.forward(x, decode_mode=1)
    prog = eng.program(x, 1)
    # launches of the 12-wave F(4x4,3x3) kernels on the large maps: conv_wino4_kernel (cfg 70) and, since round 6,
    # conv_wino4w_kernel (cfg 86: the 96-channel layers)
    n70 = sum(1 for m in prog.meta if m['kind'] == 'conv' and m['cfg'] in (70, 86))
    torch.cuda.synchronize()
    return maps.clone(), n70


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    a = ap.parse_args()
    x = synth.synth_crops(a.batch, 3, 256, 256, seed=100).cuda()
    base, n = run(x, {'EGONET_AMD_F43': '0'})
    print('F43 off: %d launches of cfg 70, max|map| %.3f' % (n, float(base.abs().max())))
    again, _ = run(x, {'EGONET_AMD_F43': '0'})
    print('F43 off, second engine: max diff %.3e' % float((again - base).abs().max()))
    for match in ('', 'ci48.48_co48.48_k3x3_s1_p1_r0', 'ci48.48_co48.48_k3x3_s1_p1_r1', 'ci96.96_co96.96_k3x3_s1_p1_r0',
                  'ci96.96_co96.96_k3x3_s1_p1_r1', 'ci256.256_co48.48', 'ci96.96_co48.48'):
        got, n = run(x, {'EGONET_AMD_F43_MATCH': match} if match else {})
        d = (got - base).abs()
        print('F43 on for %-36s %3d launches of cfg 70: max diff %.3e  mean %.3e' % (
            match or '(every shape the table has it for)', n, float(d.max()), float(d.mean())), flush=True)


if __name__ == '__main__':
    main()
