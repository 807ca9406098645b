#!/usr/bin/env bash
# Round 6, call 43: where the cell-tile scan's time goes -- per-tile stamps / counters, and rocprofv3 kernel stats of the whole pruned search.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c43; mkdir -p $OUT/trace; export TMPDIR=/tmp; ROOT=$PWD
ANNLITE_DEBUG_COUNTERS=1 timeout 600 python scripts/prof_ivf_bytes.py --probe 16 2>&1 | tail -48 | tee $OUT/ivf_tiles_p16.txt
rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -- python scripts/prof_ivf_bytes.py --probe 16 --loop 50 > $OUT/trace.log 2>&1
python - <<PY | tee $OUT/ivf_kernel_stats_p16.txt
import csv,glob
for f in glob.glob('$OUT/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if int(r['Calls']) >= 50 and int(r['Calls']) <= 60: print('%-90s calls=%-4s avg_us=%8.1f min_us=%8.1f' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
rm -rf $OUT/trace
