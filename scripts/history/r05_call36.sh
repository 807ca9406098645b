#!/usr/bin/env bash
# Round 5, GPU call 36: knobs of the byte-table scan at the shard size (1.25M rows, two streams) on the final tree: table target T,
# import cadence, ring limit, seed rows -- the consumer wave is 75-85 % busy there (call 33's counters)
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; OUT=gpurun_out/r05c36; mkdir -p $OUT
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']
    print('ms/step %.4f  q/s %.0f  kernel_ms %.4f  sha %s' % (d['ms_per_step'], d['value'], r['kernel_ms'], d['result_sha256'][:10]))
except Exception as e: print('ERR', e)
PY
}
C="--rows 1250000 --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --streams 2 --steps 200 --warmup 20"
i=0
for E in "" "ANNLITE_Q8_TARGET=112" "ANNLITE_Q8_TARGET=127" "ANNLITE_Q8_TARGET=80" "ANNLITE_Q8_TUNE=255,16,384,1" "ANNLITE_Q8_TUNE=255,16,384,7" "ANNLITE_Q8_TUNE=255,16,256,3" "ANNLITE_Q8_TUNE=255,16,448,3" "ANNLITE_SEED_ROWS=49152" "ANNLITE_SEED_ROWS=24576" ""; do
  i=$((i+1)); env $E timeout 100 python bench.py $C > $OUT/t_$i.json 2>/dev/null; echo "[$E]: $(line $OUT/t_$i.json)"
done
