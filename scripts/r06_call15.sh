#!/usr/bin/env bash
# Round 6, call 15: the pair walk with the next pair requested before the merge -- tests, C5 at 5M rows (GPU-built graph).
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c15; mkdir -p $OUT
timeout 900 python -m pytest tests/test_graph_pair.py tests/test_graph_packed.py tests/test_graph_gpu_build.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest.txt
timeout 900 python scripts/bench_hnsw.py --rows 5000000 --steps 20 --build gpu > $OUT/bench_hnsw_5m.json 2> $OUT/bench_hnsw_5m.err
tail -3 $OUT/bench_hnsw_5m.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06c15/bench_hnsw_5m.json') if l.startswith('{')][-1])
r = d['roofline']
print(' c5: %.0f q/s recall %.4f build_s %.1f' % (d['value'], d['recall_at_10'], d['build_s']))
print(' walk kernel_ms %.4f (one at a time %s, ratio %.3f) expansions/query %.1f rows/query %.1f prefetched used %.3f' % (r['kernel_ms'], r.get('one_at_a_time_kernel_ms'), r['kernel_ms'] / r['one_at_a_time_kernel_ms'], r['expansions_per_query'], r['rows_evaluated_per_query'], r['prefetched_records_used']))
print(' cycles', r['cycles_per_query_by_phase'])
PY
