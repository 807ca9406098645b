#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c12; mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s); timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - T0 )) s"; tail -4 $OUT/pytest.log
python bench.py --rows 1000000 --legs none --steps 40 --warmup 10 > $OUT/bench_c2_alone.json 2>/dev/null
python bench.py --data uniform --legs none --steps 10 --warmup 3 --cpu-queries 0 --recall-queries 32 > $OUT/bench_uniform_alone.json 2>/dev/null
python bench.py --data uniform --legs none --steps 30 --warmup 10 --cpu-queries 0 --recall-queries 32 > $OUT/bench_uniform_alone_long.json 2>/dev/null
python bench.py --legs rerank --cpu-queries 0 --steps 20 --warmup 5 > $OUT/bench_rerank.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c12/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f  %s rerank %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], r.get('kernel_choice'), d.get('rerank')))
    except Exception as e: print(f, 'ERR', e)
PY
