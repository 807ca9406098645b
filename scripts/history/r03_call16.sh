#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c16; mkdir -p $OUT
export TMPDIR=/tmp
python scripts/stress_fixture.py 6 | grep -v "^ 4\|first"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "m64 or config4 or fixture or random_shapes or ties or variants" 2>&1 | tail -2
C4="--dim 768 --m 64 --batch 256 --metric cosine --legs none --cpu-queries 0 --recall-queries 32 --steps 20 --warmup 5"
timeout 300 python bench.py --rows 10000000 $C4 > $OUT/bench_c4_10m_wd16.json 2> $OUT/err.txt
for d in 8 12 24 32; do ANNLITE_HIP_LIB=$ROOT/build_exp/lib_wd$d.so timeout 300 python bench.py --rows 10000000 $C4 > $OUT/bench_c4_10m_wd$d.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c16/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f  %s recall %.3f' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], r['kernel'], d['recall_at_10']))
    except Exception as e: print(f, 'ERR', e)
PY
