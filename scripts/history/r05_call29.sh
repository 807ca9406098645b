#!/usr/bin/env bash
# Round 5, GPU call 29: seed rows (ANNLITE_SEED_ROWS) for the other legs on one box: uniform vectors, config 4 (M = 64), the headline at k = 10
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; OUT=gpurun_out/r05c29; mkdir -p $OUT
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']
    print('ms/step %.4f  q/s %.0f  kernel_ms %.4f' % (d['ms_per_step'], d['value'], r['kernel_ms']))
except Exception as e: print('ERR', e)
PY
}
C="--legs none --cpu-queries 0 --recall-queries 0 --no-rerank --streams 2"
for S in 32768 65536 131072; do ANNLITE_SEED_ROWS=$S timeout 200 python bench.py --data uniform --steps 20 --warmup 5 $C > $OUT/uniform_$S.json 2>/dev/null; echo "uniform S=$S: $(line $OUT/uniform_$S.json)"; done
for S in 8192 16384 32768; do ANNLITE_SEED_ROWS=$S timeout 300 python bench.py --dim 768 --m 64 --batch 256 --metric cosine --steps 20 --warmup 5 $C > $OUT/c4_$S.json 2>/dev/null; echo "c4 S=$S: $(line $OUT/c4_$S.json)"; done
for S in 32768 49152 65536 24576; do ANNLITE_SEED_ROWS=$S timeout 200 python bench.py --steps 100 --warmup 10 $C > $OUT/k10_$S.json 2>/dev/null; echo "k10 10M S=$S: $(line $OUT/k10_$S.json)"; done
