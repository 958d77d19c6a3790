#!/usr/bin/env python
"""Turn two rocprofv3 --pmc passes of bench.py (FETCH_SIZE in one run,
WRITE_SIZE in another -- they do not fit one pass on gfx950) into the per-kernel
HBM traffic table bench.py reads (profiles/r1_pmc_traffic.json).

    python tools/pmc_traffic.py <dir-of-FETCH_SIZE-run> <dir-of-WRITE_SIZE-run> > profiles/r1_pmc_traffic.json

Units and corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide
(16 B/lane) coalesced read stream, so it is doubled.  All our conv / fuse /
decode kernels read with 16 B per lane.  Per kernel symbol the mean over every
dispatch of the run is stored (per launch, like roofline.achieved).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(root, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row['Counter_Name'] == counter:
                    acc[row['Kernel_Name']].append(float(row['Counter_Value']))
    return acc


def main():
    # one or more (FETCH_SIZE run, WRITE_SIZE run) directory pairs: bench.py, tools/train_hc_bench.py,
    # tools/train_bench.py -- merged into one table keyed by kernel symbol
    fetch, write = defaultdict(list), defaultdict(list)
    dirs = sys.argv[1:]
    for i in range(0, len(dirs) - 1, 2):
        for k, v in per_kernel(dirs[i], 'FETCH_SIZE').items():
            fetch[k] += v
        for k, v in per_kernel(dirs[i + 1], 'WRITE_SIZE').items():
            write[k] += v
    out = {'_note': 'HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, mean over all dispatches of '
                    'one short run per counter and program (bench.py, tools/train_hc_bench.py, tools/train_bench.py) (gfx950: FETCH_SIZE counts 64 B per '
                    '128-B request of a wide stream)', 'kernels': {}}
    for name in sorted(set(fetch) | set(write)):
        if not any(k in name for k in ('conv_', 'fuse', 'decode', 'nchw', 'nhwc', 'kpts', 'pose', 'unnorm', 'ramps', 'gemm',
                                       'bn_', 'colreduce', 'wgrad', 'adam', 'elem_loss', 'mse_')):
            continue
        f = fetch.get(name, [])
        w = write.get(name, [])
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        out['kernels'][name] = {'dispatches': max(len(f), len(w)), 'fetch_kib_raw': fm, 'write_kib': wm,
                                'hbm_bytes_per_launch': (2.0 * fm + wm) * 1024.0}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
