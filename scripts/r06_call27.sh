#!/usr/bin/env bash
# Round 6, call 27: config 5 with consecutive batches on two caller streams (and one, for the record); the graph leg likewise.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c27; mkdir -p $OUT
timeout 900 python scripts/bench_hnsw.py --rows 5000000 --steps 40 --build gpu > $OUT/bench_hnsw_5m.json 2> $OUT/bench_hnsw_5m.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06c27/bench_hnsw_5m.json') if l.startswith('{')][-1])
print(' c5: %.0f q/s (%.4f ms per batch, %d streams) recall %.4f build_s %.1f walk kernel %.4f ms' % (d['value'], d['ms_per_step'], d['streams'], d['recall_at_10'], d['build_s'], d['roofline']['kernel_ms']))
for k in d:
    if k.startswith('hnsw_') or k.startswith('exhaustive'):
        print(' ', k, d[k])
PY
timeout 600 python bench.py --legs graph --cpu-queries 0 --steps 40 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('graph leg', d['graph'])"
