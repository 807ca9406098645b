#!/usr/bin/env bash
# rocprofv3 kernel stats of the pruned (IVF) search: scripts/bench_ivf.py at one (cells, probes) point
TAG=${1:-ivf}; CELLS=${2:-256}; PROBES=${3:-16}
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.." ; ROOT=$PWD
rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -- python scripts/bench_ivf.py --cells $CELLS --probes $PROBES --no-rerank --reps 20 > $OUT/bench.log 2>&1
python - <<PY > $OUT/summary.txt
import csv,glob
print('command: rocprofv3 --kernel-trace --stats -- python scripts/bench_ivf.py --cells $CELLS --probes $PROBES --no-rerank --reps 20')
print([l for l in open('$OUT/bench.log') if l.startswith('{')][-1].strip()[:1200])
for f in glob.glob('$OUT/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'annlite' in r['Name']: print('%-100s calls=%-4s avg_us=%9.1f min_us=%9.1f max_us=%9.1f pct=%s' % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Percentage']))
PY
cat $OUT/summary.txt
