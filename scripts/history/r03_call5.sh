#!/usr/bin/env bash
# round 3, GPU call 5: per-wave candidate rings (e4) against the shared ring (e3)
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c5; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
P="python scripts/prof_scan.py --data lowrank --fused --iters 24 --valid"
Q="--legs none --cpu-queries 8 --cpu-repeats 1 --steps 60 --warmup 10"
for lib in e3 e4; do
  export ANNLITE_HIP_LIB=$ROOT/build_exp/lib_$lib.so
  $P --rows 1250000 > $OUT/plain_${lib}.txt 2>&1
  ANNLITE_DEBUG_COUNTERS=2 $P --rows 1250000 > $OUT/timeline_${lib}.txt 2>&1
  ANNLITE_DEBUG_COUNTERS=1 $P --rows 1250000 > $OUT/counters_${lib}.txt 2>&1
  for st in 1 2; do python bench.py --rows 1250000 $Q --streams $st > $OUT/bench_1p25m_${lib}_s$st.json 2>/dev/null; done
  python bench.py $Q --steps 30 > $OUT/bench_10m_${lib}.json 2>/dev/null
done
unset ANNLITE_HIP_LIB
grep -H "scan kernel\|timeline\|byte-table kernel:" $OUT/*.txt | cut -c1-400
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c5/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f  parity %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], d['cpu_baseline'] and d['cpu_baseline']['gpu_matches_cpu_bit_exact']))
    except Exception as e: print(f, 'ERR', e)
PY
