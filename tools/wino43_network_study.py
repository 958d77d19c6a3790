#!/usr/bin/env python
"""Go / no-go for Winograd F(4x4,3x3) on the C >= 96 layers of HRNet-W48 (CPU, torch fp32): the whole network with
those 3x3 stride-1 convolutions evaluated by an fp32 F(4x4,3x3) emulation (filter transform in float64, data
transforms / products / channel sums in fp32), against the fp32 oracle: heat-map error, arg-max flips, soft-arg-max.

    python tools/wino43_network_study.py [--crops 8] [--min-c 96]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonet_amd import configs, synth                     # noqa: E402
from egonet_amd.model.heatmapModel import hrnet           # noqa: E402
from oracle import hrnet_oracle, decode_oracle            # noqa: E402
from tools.wino43_error_study import mats                  # noqa: E402

AT, G, BT = [torch.tensor(m) for m in mats([0.0, 1.0, -1.0, 2.0, -2.0])]


def conv_f43(x, w):
    """x [N,C,H,W] fp32, w [Co,C,3,3]; pad 1, stride 1."""
    n, c, h, wd = x.shape
    assert h % 4 == 0 and wd % 4 == 0
    U = torch.einsum('ia,ocab,jb->ocij', G, w.double(), G).float()
    xp = F.pad(x, (1, 1, 1, 1))
    d = F.unfold(xp, kernel_size=6, stride=4).view(n, c, 6, 6, -1)              # [N,C,6,6,T]
    BTf, ATf = BT.float(), AT.float()
    V = torch.einsum('ia,ncabt,jb->ncijt', BTf, d, BTf)
    M = torch.einsum('ocij,ncijt->noijt', U, V)
    Y = torch.einsum('ia,noabt,jb->noijt', ATf, M, ATf)                           # [N,Co,4,4,T]
    co = w.shape[0]
    return F.fold(Y.reshape(n, co * 16, -1), (h, wd), kernel_size=4, stride=4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--crops', type=int, default=8)
    ap.add_argument('--min-c', type=int, default=96)
    a = ap.parse_args()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = configs.w48_config('heatmap')
    net = hrnet.get_pose_net(cfg, is_train=False)
    sd = synth.synth_state_dict(net.state_dict(), seed=1)
    x = synth.synth_crops(a.crops, 3, 256, 256, seed=100)
    ref = hrnet_oracle.hrnet_forward(sd, cfg, x).numpy()
    orig = hrnet_oracle._conv
    count = [0]

    def conv(sdict, key, xx, stride=1, pad=0):
        w = sdict[key + '.weight']
        if w.shape[2] == 3 and stride == 1 and pad == 1 and w.shape[0] >= a.min_c and w.shape[1] >= a.min_c \
                and xx.shape[2] % 4 == 0 and xx.shape[3] % 4 == 0:
            count[0] += 1
            out = conv_f43(xx, w)
            b = sdict.get(key + '.bias')
            return out if b is None else out + b.view(1, -1, 1, 1)
        return orig(sdict, key, xx, stride, pad)
    hrnet_oracle._conv = conv
    try:
        got = hrnet_oracle.hrnet_forward(sd, cfg, x).numpy()
    finally:
        hrnet_oracle._conv = orig
    print('%d convolutions on F(4x4,3x3); heat-maps span [%.1f, %.1f], rms %.2f' % (count[0], ref.min(), ref.max(), np.sqrt((ref ** 2).mean())))
    print('max |maps - oracle| = %.2e   (bar 5e-4)' % np.abs(got - ref).max())
    ia, _ = decode_oracle.argmax_index(got)
    ib, mb = decode_oracle.argmax_index(ref)
    flat = ref.reshape(ref.shape[0], ref.shape[1], -1)
    top2 = np.sort(flat, axis=2)[:, :, -2:]
    gap = top2[:, :, 1] - top2[:, :, 0]
    print('arg-max: %d of %d maps differ; smallest top-2 gap of the oracle maps %.2e, maps with gap < 5e-4: %d' % (
        int((ia != ib).sum()), ia.size, gap.min(), int((gap < 5e-4).sum())))
    sa, _ = decode_oracle.soft_arg_max(got)
    sb, _ = decode_oracle.soft_arg_max(ref)
    print('max |soft-arg-max - oracle| = %.2e px   (bar 1e-3)' % np.abs(sa - sb).max())


if __name__ == '__main__':
    main()
