#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c23; mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --rows 1250000 --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 60 --warmup 10"
$B > $OUT/bench_base.json 2>/dev/null
for im in 0 1 7; do ANNLITE_Q8_TUNE="15,16,384,$im" $B > $OUT/bench_import$im.json 2>/dev/null; done
for t in 64 80 112 127; do ANNLITE_Q8_TARGET=$t $B > $OUT/bench_target$t.json 2>/dev/null; done
for e in "3,4" "7,8" "31,32"; do ANNLITE_Q8_TUNE="$e,384,3" $B > $OUT/bench_epoch_${e/,/_}.json 2>/dev/null; done
for s in 16384 49152; do ANNLITE_SEED_ROWS=$s $B > $OUT/bench_seed$s.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c23/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))
    except Exception as e: print(f, 'ERR', e)
PY
