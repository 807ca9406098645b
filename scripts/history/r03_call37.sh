#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c37; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not config5 and not bench_under" 2>&1 | tail -5
python scripts/stress_fixture.py 4 | grep -v "^ 4\|first" | tail -4
for rows in 1250000 10000000; do
  P="python scripts/prof_scan.py --data lowrank --fused --valid --rows $rows --iters 12"
  echo "== $rows"; $P 2>/dev/null | grep "scan kernel ms" | cut -c1-90
  ANNLITE_DEBUG_COUNTERS=2 $P 2>/dev/null | grep "timeline" | cut -c60-330
  ANNLITE_DEBUG_COUNTERS=1 $P 2>/dev/null | grep "byte-table kernel: wave" | cut -c1-260
done
B="python bench.py --legs none --cpu-queries 0 --recall-queries 0 --no-rerank"
$B --rows 1250000 --streams 2 > $OUT/bench_1p25m_s2.json 2>/dev/null
$B --rows 1250000 > $OUT/bench_1p25m_s1.json 2>/dev/null
$B > $OUT/bench_10m.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c37/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))
PY
