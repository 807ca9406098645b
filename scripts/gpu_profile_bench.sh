#!/usr/bin/env bash
# rocprofv3 kernel stats of bench.py itself (the command whose JSON line is reported); extra args go to bench.py
TAG=${1:-bench}; shift || true
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.." ; ROOT=$PWD
rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -- python bench.py --no-rerank --recall-queries 0 --cpu-queries 0 "$@" > $OUT/bench.log 2>&1
python - <<PY > $OUT/summary.txt
import csv,glob
print('command: rocprofv3 --kernel-trace --stats -- python bench.py --no-rerank --recall-queries 0 --cpu-queries 0 $@')
print(open('$OUT/bench.log').read().strip().splitlines()[-3 if False else -1][:400] if False else [l for l in open('$OUT/bench.log') if l.startswith('{')][-1].strip())
for f in glob.glob('$OUT/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage']) > 0.01: print('%-80s calls=%-4s avg_us=%9.1f min_us=%9.1f max_us=%9.1f pct=%s' % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Percentage']))
PY
cat $OUT/summary.txt
