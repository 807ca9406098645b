#!/usr/bin/env python3
"""Condense a scripts/gpu_profile.sh output directory into one text summary (per kernel: calls, avg
duration from the kernel trace stats; per-dispatch average of every PMC counter)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
print('== kernel stats (rocprofv3 --kernel-trace --stats) ==')
for f in glob.glob(os.path.join(root, 'trace', '**', '*kernel_stats.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        print('%-90s calls=%-4s avg_ns=%-12s total_ns=%-14s pct=%s' % (r['Name'][:90], r['Calls'], r['AverageNs'], r['TotalDurationNs'], r['Percentage']))
print(open(os.path.join(root, 'trace.log')).read().strip().splitlines()[-1])
for tag in ('pmc_a', 'pmc_b', 'pmc_c', 'pmc_d'):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, tag, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    print('== %s: per-dispatch mean ==' % tag)
    for kname, ctrs in acc.items():
        if 'adc_scan' not in kname and 'lut_' not in kname and 'merge' not in kname:
            continue
        print(' ', kname)
        for c, v in sorted(ctrs.items()):
            print('      %-28s %.4g  (n=%d)' % (c, sum(v) / len(v), len(v)))
