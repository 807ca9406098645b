#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c20; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "uint16 or code16 or wide_codes" 2>&1 | tail -15
for ks in 512; do
  timeout 300 python bench.py --rows 2000000 --m 8 --ks $ks --legs none --cpu-queries 8 --cpu-repeats 1 --recall-queries 32 --no-rerank > $OUT/bench_code16_ks$ks.json 2>$OUT/err_$ks.txt || tail -5 $OUT/err_$ks.txt
  ANNLITE_SCAN_VARIANT=31 timeout 300 python bench.py --rows 2000000 --m 8 --ks $ks --legs none --cpu-queries 0 --recall-queries 32 --no-rerank > $OUT/bench_code16_ks${ks}_u16.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c20/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f  %s recall %.3f parity %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], r['kernel'], d['recall_at_10'], d['cpu_baseline'] and d['cpu_baseline']['gpu_matches_cpu_bit_exact']))
    except Exception as e: print(f, 'ERR', e)
PY
