#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c41; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fixture or random_shapes or ties or bench_distribution or uint16 or rerank or candidate" 2>&1 | tail -3
python scripts/stress_fixture.py 3 | grep -v "^ 4\|first" | tail -2
B="python bench.py --legs none --cpu-queries 8 --cpu-repeats 1 --recall-queries 32 --no-rerank"
$B --layout plain > $OUT/bench_10m_plain.json 2>/dev/null
$B --layout skewed > $OUT/bench_10m_skewed.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c41/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f parity %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], d['cpu_baseline']['gpu_matches_cpu_bit_exact']))
PY
