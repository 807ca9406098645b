#!/usr/bin/env bash
# round 3, GPU call 1: parity of the reworked byte-table kernel + A/B against round 2's library at shard size
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c1; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
Q="--no-rerank --ivf-cells 0 --cpu-queries 16 --cpu-repeats 1 --steps 60 --warmup 10"
for lib in r02 e1; do
  export ANNLITE_HIP_LIB=$ROOT/build_exp/lib_$lib.so
  for st in 1 2; do
    timeout 300 python bench.py --rows 1250000 $Q --streams $st > $OUT/bench_1p25m_${lib}_s$st.json 2> $OUT/bench_1p25m_${lib}_s$st.err
  done
  timeout 300 python bench.py $Q --steps 30 > $OUT/bench_10m_${lib}.json 2> $OUT/bench_10m_${lib}.err
  ANNLITE_DEBUG_COUNTERS=1 timeout 200 python scripts/prof_scan.py --rows 1250000 --data lowrank --fused --valid > $OUT/counters_1p25m_$lib.txt 2>&1
  ANNLITE_DEBUG_SKIP=4 timeout 200 python scripts/prof_scan.py --rows 1250000 --data lowrank --fused --valid > $OUT/skip4_1p25m_$lib.txt 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c1/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f  parity %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], d['cpu_baseline'] and d['cpu_baseline']['gpu_matches_cpu_bit_exact']))
    except Exception as e: print(f, 'ERR', e)
PY
grep -h "byte-table\|scan kernel" $OUT/counters_*.txt $OUT/skip4_*.txt
