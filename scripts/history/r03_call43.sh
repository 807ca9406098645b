#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c43; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_gpu.txt
B="python bench.py --legs none --cpu-queries 0 --recall-queries 32 --no-rerank"
$B > $OUT/bench_10m.json 2>/dev/null
$B --rows 1250000 --streams 2 > $OUT/bench_1p25m_s2.json 2>/dev/null
$B --rows 2000000 --m 8 --ks 768 > $OUT/bench_m8_ks768_2m.json 2>/dev/null
$B --rows 2000000 --m 8 --ks 512 > $OUT/bench_m8_ks512_2m.json 2>/dev/null
$B --rows 10000000 --dim 64 --m 8 > $OUT/bench_m8_u8_10m.json 2>/dev/null
$B --rows 10000000 --dim 768 --m 64 --batch 256 --metric cosine > $OUT/bench_c4_10m.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c43/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))
PY
