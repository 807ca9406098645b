#!/usr/bin/env bash
# Round 6, call 48: the plan with its pairs in registers, the seed rows evaluated for their own query only; tests, A/B, kernel stats.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c48; mkdir -p $OUT/trace; export TMPDIR=/tmp; ROOT=$PWD
timeout 600 python -m pytest tests/test_ivf_byte_tiles.py tests/test_ivf.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest_ivf.txt
run() { echo "$1" | tee -a $OUT/ivf_knobs.txt; env $1 timeout 600 python scripts/bench_ivf_bytes.py --probes ${2:-16} 2>&1 | grep n_probe | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['n_probe'], r['bytes'], r['paths_bit_equal'], r['stages'])" | tee -a $OUT/ivf_knobs.txt; }
run ANNLITE_X=0 8,16,32
run ANNLITE_SEED_ROWS=131072
run ANNLITE_SEED_ROWS=32768
rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -- python scripts/prof_ivf_bytes.py --probe 16 --loop 50 > $OUT/trace.log 2>&1
python - <<PY | tee $OUT/ivf_kernel_stats_p16.txt
import csv,glob
for f in glob.glob('$OUT/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if int(r['Calls']) >= 50 and int(r['Calls']) <= 60: print('%-90s calls=%-4s avg_us=%8.1f min_us=%8.1f' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
rm -rf $OUT/trace
