#!/usr/bin/env python
"""Re-measure the F(4x4,3x3) configurations of conv_wino4.hip (cfg 70 / 80 / 82 / 83 / 84) against each table entry's best
other configuration, for every shape of `tuned/gfx950.json` one of them plans for, and rewrite the entries.

Round 4 changed what these kernels cost after the table was measured: conv_wino4c_kernel (cfg 82 / 83: four 8 x 8
images per region, K split) is new, and the item order of all of them puts co-tiles on the XCD axis from 4 co-tiles up.
The tuner now times a configuration as a one-op program (tuner._time_cfg), i.e. the K split as ONE launch with ticket
words, the way the engine runs it.

    python tools/retune_f43.py --out gpurun_out/gfx950.json [--rounds 3]
"""
import argparse
import ctypes as C
import json
import os
import re
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egonet_amd import _lib, tuner  # noqa: E402

F43 = (70, 80, 82, 83, 84)
KEY = re.compile(r'n(\d+)_h(\d+)_w(\d+)_ci(\d+)\.(\d+)_co(\d+)\.(\d+)_k(\d+)x(\d+)_s(\d+)_p(\d+)_r(\d+)_o(\d+)$')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--rounds', type=int, default=3)
    a = ap.parse_args()
    L = _lib.lib()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    table = dict(tuner._load())
    out = (C.c_int * 12)()
    changed = 0
    for key in sorted(table):
        m = KEY.match(key)
        if not m:
            continue
        v = [int(g) for g in m.groups()]
        args = tuple(v[:11]) + (bool(v[11]), bool(v[12]))
        cands = [c for c in F43 if L.egn_conv_config_kind(c) == 3 and
                 L.egn_conv_plan_query(*v[:11], v[12], c, out) == 0]
        if not cands:
            continue
        entry = table[key]
        ms = {int(k): float(t) for k, t in entry.get('ms', {}).items()}
        others = {c: t for c, t in ms.items() if c not in F43 and L.egn_conv_config_kind(c) >= 0}
        ref = min(others, key=others.get) if others else 0
        only = set(cands) | ({ref} if ref else set())
        best = {}
        for _ in range(a.rounds):            # (rounds interleave the candidates: clock state is shared)
            _, times = tuner.tune(dev, args, only=only)
            for c, t in times.items():
                best[c] = min(t, best.get(c, t))
        ms.update(best)
        pick = min((c for c in ms if L.egn_conv_config_kind(c) >= 0), key=lambda c: ms[c])
        if pick != entry['cfg']:
            changed += 1
        print('%-58s %2d -> %2d   %s' % (key, entry['cfg'], pick, '  '.join('%d: %.1f us' % (c, 1e3 * best[c]) for c in sorted(best))),
              flush=True)
        entry['cfg'] = pick
        entry['ms'] = {str(c): round(t, 5) for c, t in sorted(ms.items())}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, 'w') as f:
        json.dump(table, f, indent=0, sort_keys=True)
    print('%d entries, %d picks changed -> %s' % (len(table), changed, a.out))


if __name__ == '__main__':
    main()
