#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c42; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for v in base sd16; do
  [ $v = base ] && unset ANNLITE_HIP_LIB || export ANNLITE_HIP_LIB=$ROOT/build_exp/lib_$v.so
  P="python scripts/prof_scan.py --data lowrank --fused --valid --rows 1250000 --iters 12"
  echo "== $v"; ANNLITE_DEBUG_COUNTERS=2 $P 2>/dev/null | grep "timeline" | cut -c60-330
  ANNLITE_DEBUG_COUNTERS=1 $P 2>/dev/null | grep "byte-table kernel: wave" | cut -c1-260
  python bench.py --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --rows 1250000 --streams 2 > $OUT/bench_1p25m_s2_$v.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c42/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))
PY
