#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c32; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "not config5 and not bench_under" 2>&1 | tail -6
for v in 0 31; do
  ANNLITE_SCAN_VARIANT=$v timeout 300 python bench.py --rows 2000000 --dim 64 --m 8 --legs none --cpu-queries 8 --cpu-repeats 1 --recall-queries 32 --no-rerank > $OUT/bench_m8_2m_v$v.json 2>$OUT/err_$v.txt || tail -3 $OUT/err_$v.txt
  ANNLITE_SCAN_VARIANT=$v timeout 300 python bench.py --rows 10000000 --dim 64 --m 8 --legs none --cpu-queries 0 --recall-queries 32 --no-rerank > $OUT/bench_m8_10m_v$v.json 2>/dev/null
done
timeout 300 python bench.py --rows 10000000 --dim 64 --m 8 --legs none --cpu-queries 0 --recall-queries 32 --no-rerank --layout plain > $OUT/bench_m8_10m_plain.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c32/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f  %s (%s) recall %.3f parity %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], r['kernel'], r.get('kernel_choice'), d['recall_at_10'], d['cpu_baseline'] and d['cpu_baseline']['gpu_matches_cpu_bit_exact']))
    except Exception as e: print(f, 'ERR', e)
PY
