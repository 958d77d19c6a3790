"""Top up the shipped tile-configuration table (egonet_amd/tuned/gfx950.json) after new kernel
configurations were added: for every shape in the table, time the configurations that plan for it
but have no measurement yet, merge, re-pick the fastest.

    python tools/retune.py --out gpurun_out/gfx950.json [--match k3x3_s1]
"""
import argparse
import json
import os
import re
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from egonet_amd import tuner                                          # noqa: E402

KEY = re.compile(r'n(\d+)_h(\d+)_w(\d+)_ci(\d+)\.(\d+)_co(\d+)\.(\d+)_k(\d+)x(\d+)_s(\d+)_p(\d+)_r(\d+)_o(\d+)$')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--match', default='')
    ap.add_argument('--retime', default='', help='comma list of cfg ids to measure again in this run (same box, same '
                    'session as the new ones: boxes of the pool differ by a few per cent)')
    a = ap.parse_args()
    with open(tuner.TABLE_PATH) as f:
        table = json.load(f)
    dev = torch.device('cuda:0')
    changed = 0
    for key in sorted(table):
        m = KEY.match(key)
        if not m or not all(part in key for part in a.match.split(',')):      # (--match a,b: every part)
            continue
        args = tuple(int(v) for v in m.groups())
        again = {int(v) for v in a.retime.split(',') if v}
        have = {int(k) for k in table[key].get('ms', {})} - again
        _, times = tuner.tune(dev, args, skip=have)
        if not times:
            continue
        ms = dict(table[key].get('ms', {}))
        ms.update({str(k): round(v, 5) for k, v in times.items()})
        best = int(min(ms, key=ms.get))
        if best != int(table[key]['cfg']):
            print('%s: cfg %s (%.4f ms) -> %d (%.4f ms)' % (key, table[key]['cfg'], ms[str(table[key]['cfg'])],
                                                           best, ms[str(best)]))
            changed += 1
        table[key] = {'cfg': best, 'ms': ms}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, 'w') as f:
        json.dump(table, f, indent=0, sort_keys=True)
    print('%d entries changed' % changed)


if __name__ == '__main__':
    main()
