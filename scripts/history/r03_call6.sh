#!/usr/bin/env bash
# round 3, GPU call 6: one-launch preparation (tables + parameters + reset + seed) and per-item timeline
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c6; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
P="python scripts/prof_scan.py --data lowrank --fused --iters 24 --valid"
Q="--legs none --cpu-queries 8 --cpu-repeats 1 --steps 60 --warmup 10"
ANNLITE_DEBUG_COUNTERS=2 $P --rows 1250000 > $OUT/items_1p25m.txt 2>&1
ANNLITE_DEBUG_COUNTERS=2 ANNLITE_DEBUG_SKIP=4 $P --rows 1250000 > $OUT/items_1p25m_skip4.txt 2>&1
ANNLITE_DEBUG_COUNTERS=2 $P --rows 10000000 --iters 10 > $OUT/items_10m.txt 2>&1
grep -H "scan kernel\|timeline\|items:\|step loop by" $OUT/items*.txt | cut -c1-330
for fs in 0 1; do
  if [ $fs = 1 ]; then export ANNLITE_NO_FUSED_SEED=1; else unset ANNLITE_NO_FUSED_SEED; fi
  for st in 1 2; do python bench.py --rows 1250000 $Q --streams $st > $OUT/bench_1p25m_nofuse${fs}_s$st.json 2>/dev/null; done
  python bench.py $Q --steps 30 > $OUT/bench_10m_nofuse${fs}.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c6/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f  parity %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], d['cpu_baseline'] and d['cpu_baseline']['gpu_matches_cpu_bit_exact']))
    except Exception as e: print(f, 'ERR', e)
PY
