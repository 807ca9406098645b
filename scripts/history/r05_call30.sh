#!/usr/bin/env bash
# Round 5, GPU call 30: epoch schedule of the byte-table scan (ANNLITE_Q8_TUNE=epoch0,mul,ring,import) at the small tables and at 10M rows, one box, two streams
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; OUT=gpurun_out/r05c30; mkdir -p $OUT
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']
    print('ms/step %.4f  q/s %.0f  kernel_ms %.4f  sha %s' % (d['ms_per_step'], d['value'], r['kernel_ms'], d['result_sha256'][:10]))
except Exception as e: print('ERR', e)
PY
}
C="--legs none --cpu-queries 0 --recall-queries 0 --no-rerank --streams 2 --steps 200 --warmup 20"
for rows in 1250000 1000000 10000000; do
  for T in "15,16,384,3" "63,16,384,3" "255,16,384,3" "100000,16,384,3" "15,16,384,3"; do
    ANNLITE_Q8_TUNE=$T timeout 200 python bench.py --rows $rows $C > $OUT/t_${rows}_${T//,/_}.json 2>/dev/null; echo "rows $rows tune $T: $(line $OUT/t_${rows}_${T//,/_}.json)"
  done
done
