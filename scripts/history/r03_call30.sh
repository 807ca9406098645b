#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c30; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not config5 and not bench_under" 2>&1 | tail -3
python scripts/stress_fixture.py 4 | grep -v "^ 4\|first" | tail -4
B="python bench.py --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 60 --warmup 10"
$B --rows 1250000 > $OUT/bench_1p25m.json 2>/dev/null
$B --rows 1250000 --streams 2 > $OUT/bench_1p25m_s2.json 2>/dev/null
$B > $OUT/bench_10m.json 2>/dev/null
ANNLITE_DEBUG_COUNTERS=2 python scripts/prof_scan.py --data lowrank --fused --valid --rows 1250000 --iters 12 2>/dev/null | grep "timeline\|items:\|merge phase" | cut -c1-300
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c30/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))
PY
