#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c17; mkdir -p $OUT
export TMPDIR=/tmp
C4="--dim 768 --m 64 --batch 256 --metric cosine --legs none --cpu-queries 0 --recall-queries 32 --steps 20 --warmup 5"
for d in exp3 exp4; do ANNLITE_HIP_LIB=$ROOT/build_exp/lib_$d.so timeout 300 python bench.py --rows 10000000 $C4 > $OUT/bench_c4_10m_$d.json 2>$OUT/err_$d.txt; tail -2 $OUT/err_$d.txt; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c17/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f  %s recall %.3f' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], r['kernel'], d['recall_at_10']))
    except Exception as e: print(f, 'ERR', e)
PY
