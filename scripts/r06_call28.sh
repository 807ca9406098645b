#!/usr/bin/env bash
# Round 6, call 28: config 5 on 1 / 2 / 3 / 4 caller streams.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c28; mkdir -p $OUT
for s in 2 3 4; do
timeout 600 python scripts/bench_hnsw.py --rows 5000000 --steps 48 --build gpu --streams $s 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['hnsw_gpu_walk_exact_rerank']
print('streams $s: c5 %.0f q/s (%.4f ms per batch; one stream %.0f) recall %.4f; adc ranking %.0f' % (d['value'], d['ms_per_step'], r['one_stream_queries_per_s'], d['recall_at_10'], d['hnsw_gpu_walk_adc']['queries_per_s']))"
done | tee $OUT/c5_streams.txt
