#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c39; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rerank or candidate or superset or fixture or ties or bench_distribution" 2>&1 | tail -3
timeout 300 python bench.py --legs rerank --cpu-queries 0 > $OUT/bench_rerank.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03c39/bench_rerank.json')); r=d['roofline']
print('ms/step %.4f kernel %.4f q/s %.0f rerank %s' % (d['ms_per_step'], r['kernel_ms'], d['value'], d['rerank']))
PY
