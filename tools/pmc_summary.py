#!/usr/bin/env python
"""Condense a rocprofv3 counter_collection CSV into one row per
(kernel, grid) with the mean of every counter.  Usage:
    python tools/pmc_summary.py <dir-with-csv> > summary.txt
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)
    agg = defaultdict(lambda: defaultdict(list))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get('Kernel_Name', '')
                if 'conv_' not in name and 'fuse' not in name and 'decode' not in name:
                    continue
                key = (name[:60], row.get('Grid_Size', ''), row.get('LDS_Block_Size', ''), row.get('VGPR_Count', ''),
                       row.get('Accum_VGPR_Count', ''), row.get('Scratch_Size', ''))
                agg[key][row['Counter_Name']].append(float(row['Counter_Value']))
    for key, ctrs in sorted(agg.items()):
        print('%s grid=%s lds=%s vgpr=%s agpr=%s scratch=%s' % key)
        for c, v in sorted(ctrs.items()):
            print('    %-32s n=%-4d mean=%.4g' % (c, len(v), sum(v) / len(v)))


if __name__ == '__main__':
    main()
