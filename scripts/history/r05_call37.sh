#!/usr/bin/env bash
# Round 5, GPU call 37: table target T of the byte-table scan by table size (call 36: T = 80 beat 96 at 1.25M rows; 112 / 127 lose)
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; OUT=gpurun_out/r05c37; mkdir -p $OUT
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']
    print('ms/step %.4f  q/s %.0f  kernel_ms %.4f  sha %s' % (d['ms_per_step'], d['value'], r['kernel_ms'], d['result_sha256'][:10]))
except Exception as e: print('ERR', e)
PY
}
C="--legs none --cpu-queries 0 --recall-queries 0 --no-rerank --streams 2 --steps 200 --warmup 20"
run() { ANNLITE_Q8_TARGET=$2 timeout 150 python bench.py --rows $1 $C > $OUT/t_$1_$2.json 2>/dev/null; echo "rows $1 T $2: $(line $OUT/t_$1_$2.json)"; }
for T in 96 64 72 80 88 96; do run 1250000 $T; done
for T in 96 72 80; do run 1000000 $T; done
for T in 96 80 88 96; do run 10000000 $T; done
