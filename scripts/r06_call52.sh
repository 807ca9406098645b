#!/usr/bin/env bash
# Round 6, call 52: cell tiles -- the first items of an XCD's workgroups consecutive (same-cell tiles share an L2): time and FETCH_SIZE, A/B.
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=gpurun_out/r06c52; mkdir -p $OUT; export TMPDIR=/tmp
for v in 1 0 1 0; do
  echo "ANNLITE_Q8_MAP=$v" | tee -a $OUT/ivf_xcd_first.txt
  ANNLITE_Q8_MAP=$v timeout 600 python scripts/bench_ivf_bytes.py --probes 16 --reps 30 2>&1 | grep n_probe | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['n_probe'], r['bytes'], r['paths_bit_equal'], r['stages'])" | tee -a $OUT/ivf_xcd_first.txt
done
for v in 1 0; do
  ANNLITE_Q8_MAP=$v rocprofv3 --kernel-trace --kernel-include-regex "adc_scan_q8" --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $ROOT/$OUT/pmc_$v -- python scripts/prof_ivf_bytes.py --probe 16 --loop 20 > $OUT/pmc_$v.log 2>&1
  python - <<PY | tee -a $OUT/ivf_xcd_first.txt
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('$OUT/pmc_$v/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'adc_scan_q8' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('ANNLITE_Q8_MAP=$v', {c: round(sum(v)/len(v), 1) for c, v in acc.items()}, 'dispatches', len(acc.get('FETCH_SIZE', [])))
PY
  rm -rf $OUT/pmc_$v
done
timeout 600 python bench.py --legs ivf --cpu-queries 0 2>/dev/null | tail -c 400 | tee $OUT/bench_ivf_leg_tail.txt
