#!/usr/bin/env bash
# Round 5, GPU call 33: where the time goes on a table in cluster order (work items by tile / slice, debug counters) against the i.i.d. table
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; OUT=gpurun_out/r05c33; mkdir -p $OUT
for rows in 1250000 10000000; do
  for order in sorted iid; do
    ANNLITE_DEBUG_COUNTERS=1 timeout 200 python scripts/prof_scan.py --rows $rows --fused --data lowrank --order $order --iters 12 > $OUT/dbg_${rows}_$order.txt 2>&1
    echo "== rows $rows $order"; grep -v "^/opt\|warn" $OUT/dbg_${rows}_$order.txt | cut -c1-420
  done
done
