#!/usr/bin/env bash
# kernel-trace stats only: scripts/gpu_kstats.sh <tag> [prof_scan args]
TAG=${1:-k}; shift || true
OUT=gpurun_out/ks_$TAG; mkdir -p $OUT/trace; export TMPDIR=/tmp
cd "$(dirname "$0")/.." ; ROOT=$PWD
rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -- python scripts/prof_scan.py "$@" > $OUT/trace.log 2>&1
python - <<PY
import csv,glob
for f in glob.glob('$OUT/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage']) > 0.05: print('%-70s calls=%-4s avg_us=%8.1f pct=%s' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
tail -1 $OUT/trace.log
