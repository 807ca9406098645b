#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c29; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
P="python scripts/prof_scan.py --data lowrank --fused --valid --rows 1250000 --iters 24"
for s in 32768 8192 1024; do
  ANNLITE_SEED_ROWS=$s timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/s$s -- $P > $OUT/s$s.log 2>&1
  f=$(find $OUT/s$s -name '*kernel_stats.csv' | head -1)
  echo "== seed rows $s"; python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'seed_bound' in r['Name'] or 'adc_scan_q8' in r['Name']: print(r['Name'][:60], 'calls', r['Calls'], 'avg_us %.1f' % (float(r['AverageNs'])/1e3), 'min_us %.1f' % (float(r['MinNs'])/1e3))
"
done
rm -rf $OUT/s*/
