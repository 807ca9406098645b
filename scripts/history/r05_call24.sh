#!/usr/bin/env bash
# Round 5, GPU call 24: small tables with TWO batches side by side: a 4-slice plan (128 work items = half the CUs per launch) on 2 / 3 / 4 caller
# streams against the planned 8 slices (one work item per CU, whole chip per launch), c2 (1M rows) and the shard of 8 (1.25M rows); one box.
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; OUT=gpurun_out/r05c24; mkdir -p $OUT
for rows in 1000000 1250000; do
  A="--rows $rows --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 200 --warmup 20"
  for cfg in "8 2" "4 2" "4 3" "4 4" "8 2" "4 2"; do set -- $cfg; sl=$1; st=$2
    ANNLITE_SCAN_SLICES=$sl timeout 120 python bench.py $A --streams $st > $OUT/b_${rows}_sl${sl}_st${st}.json 2>/dev/null
    python - <<PY
import json
d=json.loads([l for l in open('$OUT/b_${rows}_sl${sl}_st${st}.json') if l.startswith('{')][-1]); r=d['roofline']
print('rows $rows slices $sl streams $st: ms/step %.4f  q/s %.0f  kernel_ms %.4f  sha %s' % (d['ms_per_step'], d['value'], r['kernel_ms'], d['result_sha256'][:12]))
PY
  done
done
