#!/usr/bin/env bash
# Round 6, call 30: the graph path for M = 64 (tests) and a number at config 4's shape (2M x 768, m = 64, batch 256; GPU-built graph).
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c30; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_graph_pair.py tests/test_graph_gpu_build.py tests/test_graph_packed.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest.txt
timeout 900 python scripts/bench_hnsw.py --rows 2000000 --dim 768 --m 64 --batch 256 --steps 20 --build gpu > $OUT/bench_hnsw_2m_768_m64.json 2> $OUT/bench_hnsw_2m_768_m64.err
tail -2 $OUT/bench_hnsw_2m_768_m64.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06c30/bench_hnsw_2m_768_m64.json') if l.startswith('{')][-1])
print(d['config'])
print(' %.0f q/s (%.4f ms per batch, %d streams) recall %.4f build_s %.1f walk kernel %.4f ms' % (d['value'], d['ms_per_step'], d['streams'], d['recall_at_10'], d['build_s'], d['roofline']['kernel_ms']))
for k in d:
    if k.startswith('hnsw_') or k.startswith('exhaustive'):
        print(' ', k, d[k])
PY
