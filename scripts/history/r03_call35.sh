#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c35; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "2 20" "10 60" "20 200"; do
  set -- $cfg; W=$1; K=$2
  for pw in 0 64; do
    A="--rows 1250000 --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps $K --warmup $W --prewarm-steps $pw"
    timeout 300 python bench.py $A --streams 2 > $OUT/bench_W${W}_K${K}_pw$pw.json 2>/dev/null
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c35/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f q/s %.0f' % (d['ms_per_step'], r['kernel_ms'], d['value']))
PY
