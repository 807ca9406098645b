#!/usr/bin/env bash
# round 3, GPU call 13: M = 64 byte-table kernel -- parity, then config 4 against the u16-table kernel
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c14; mkdir -p $OUT
export TMPDIR=/tmp
python scripts/stress_fixture.py 30 | grep -v "^ 4\|first"
C4="--dim 768 --m 64 --batch 256 --metric cosine --legs none --cpu-queries 8 --cpu-repeats 1 --recall-queries 32 --steps 20 --warmup 5"
for v in 0 31; do
  ANNLITE_SCAN_VARIANT=$v timeout 300 python bench.py --rows 2000000 $C4 > $OUT/bench_c4_2m_v$v.json 2> $OUT/bench_c4_2m_v$v.err
done
timeout 300 python bench.py --rows 10000000 $C4 > $OUT/bench_c4_10m.json 2> $OUT/bench_c4_10m.err
ANNLITE_DEBUG_COUNTERS=1 timeout 300 python bench.py --rows 2000000 $C4 --steps 3 --warmup 1 > /dev/null 2> $OUT/counters_c4_2m.err; tail -3 $OUT/counters_c4_2m.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c14/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f  %s recall %.3f parity %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], r['kernel'], d['recall_at_10'], d['cpu_baseline'] and d['cpu_baseline']['gpu_matches_cpu_bit_exact']))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-500:])
PY
T0=$(date +%s); timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - T0 )) s"; tail -4 $OUT/pytest.log
