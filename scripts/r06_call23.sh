#!/usr/bin/env bash
# Round 6, call 23: the EUCLIDEAN ADC ranking straight from the walk's list; config-5 tests; C5 at 5M rows (GPU-built graph).
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c23; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_graph_gpu_build.py tests/test_graph_packed.py -x -q -m gpu -k "hnsw or config5 or graph or facade" 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 900 python scripts/bench_hnsw.py --rows 5000000 --steps 20 --build gpu > $OUT/bench_hnsw_5m.json 2> $OUT/bench_hnsw_5m.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06c23/bench_hnsw_5m.json') if l.startswith('{')][-1])
print(' c5: %.0f q/s recall %.4f build_s %.1f' % (d['value'], d['recall_at_10'], d['build_s']))
for k in d:
    if k.startswith('hnsw_') or k.startswith('exhaustive'):
        print(' ', k, d[k])
PY
