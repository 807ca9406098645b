#!/usr/bin/env bash
# Round 6, call 36: the candidate generator behind the table-wide seed -- row-queue kernel variant, seed rows.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c36; mkdir -p $OUT
for v in "base" "ANNLITE_CAND_RQ=1" "ANNLITE_SEED_ROWS=16384" "ANNLITE_SEED_ROWS=65536" "base" "ANNLITE_CAND_RQ=1"; do
  E=""; [ "$v" != base ] && E="$v"
  env $E timeout 400 python bench.py --legs rerank --cpu-queries 0 --recall-queries 256 --steps 40 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['rerank']
print('%-24s main %.0f q/s; rerank leg %9.0f q/s at recall@10 %.4f' % ('$v', d['value'], r['value'], r['recall_at_10']))"
done | tee $OUT/cand_knobs.txt
