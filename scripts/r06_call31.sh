#!/usr/bin/env bash
# Round 6, call 31: M = 64 at 500k x 768 -- the GPU-built graph against the host-built one under the same walk.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c31; mkdir -p $OUT
timeout 1500 python scripts/bench_hnsw.py --rows 500000 --dim 768 --m 64 --batch 256 --steps 10 --build both > $OUT/bench_hnsw_500k_768_m64.json 2> $OUT/err.txt
tail -2 $OUT/err.txt
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06c31/bench_hnsw_500k_768_m64.json') if l.startswith('{')][-1])
print(d['config'])
print(' %.0f q/s recall %.4f build_s gpu %.1f host %.1f walk kernel %.4f ms' % (d['value'], d['recall_at_10'], d['gpu_build_s'], d['host_build_s'], d['roofline']['kernel_ms']))
for k in d:
    if k.startswith('hnsw_') or k.startswith('exhaustive'):
        print(' ', k, d[k])
PY
