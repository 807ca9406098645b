#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c24; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "library_picks or grey_zone or bench_distribution or multi_gpu or variants" 2>&1 | tail -15
cat gpurun_out/kernel_choice_2m_uniform_vectors.txt gpurun_out/kernel_choice_2m_uniform_codes.txt
timeout 300 python bench.py --data uniform --legs none --cpu-queries 0 --no-rerank > $OUT/bench_uniform.json 2>/dev/null
timeout 300 python bench.py --legs none --cpu-queries 0 --no-rerank > $OUT/bench_lowrank.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c24/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f  %s (%s) recall %.3f' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], r['kernel'], r.get('kernel_choice'), d['recall_at_10']))
    except Exception as e: print(f, 'ERR', e)
PY
