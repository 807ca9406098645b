#!/usr/bin/env bash
# Round 6, call 12: the level-0 graph built on the GPU -- kernel tests, index tests, C5 at 5M rows with both builds side by side.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c12; mkdir -p $OUT
timeout 900 python -m pytest tests/test_graph_gpu_build.py -x -q -m gpu 2>&1 | tail -25 | tee $OUT/pytest_build.txt
timeout 300 python scripts/bench_hnsw.py --rows 1000000 --steps 5 --build gpu > $OUT/bench_hnsw_1m_gpu.json 2> $OUT/bench_hnsw_1m_gpu.err
tail -3 $OUT/bench_hnsw_1m_gpu.err
timeout 1200 python scripts/bench_hnsw.py --rows 5000000 --steps 5 > $OUT/bench_hnsw_5m.json 2> $OUT/bench_hnsw_5m.err
tail -3 $OUT/bench_hnsw_5m.err
python - <<'PY'
import json
for f in ('gpurun_out/r06c12/bench_hnsw_1m_gpu.json', 'gpurun_out/r06c12/bench_hnsw_5m.json'):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'no line', e); continue
    r = d['roofline']
    print(f)
    print(' c5: %.0f q/s recall %.4f built on %s: build_s %.1f (gpu %s host %s)' % (d['value'], d['recall_at_10'], d['graph_built_on'], d['build_s'], d['gpu_build_s'], d['host_build_s']))
    print(' walk kernel_ms %.4f (one at a time %s) expansions/query %.1f rows/query %.1f' % (r['kernel_ms'], r.get('one_at_a_time_kernel_ms'), r['expansions_per_query'], r['rows_evaluated_per_query']))
    for k in d:
        if k.startswith('hnsw_') or k.startswith('exhaustive'):
            print(' ', k, d[k])
PY
