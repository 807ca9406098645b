#!/usr/bin/env bash
# Round 5, GPU call 5: graph walk with the rank-count merge + LDS-addressed visited table; k64 tests with the scaled guard.
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r05c5; mkdir -p $OUT
timeout 300 python -m pytest tests/test_graph_packed.py -x -q > $OUT/pytest_graph_packed.txt 2>&1; echo "graph_packed rc=$?"; tail -3 $OUT/pytest_graph_packed.txt
timeout 600 python -m pytest tests/test_k64_byte_tables.py -x -q > $OUT/pytest_k64.txt 2>&1; echo "k64 rc=$?"; tail -3 $OUT/pytest_k64.txt
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5 or hnsw or graph" > $OUT/pytest_graph_cases.txt 2>&1; echo "graph cases rc=$?"; tail -3 $OUT/pytest_graph_cases.txt
timeout 600 python scripts/bench_hnsw.py --rows 5000000 --steps 5 > $OUT/bench_hnsw_5m.json 2>$OUT/bench_hnsw_5m.err; echo "bench_hnsw rc=$?"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/r05c5/bench_hnsw_5m.json') if l.startswith('{')][-1])
    r = d['roofline']
    print('c5: %.0f q/s recall %.3f build %.1f s; walk q/s %s' % (d['value'], d['recall_at_10'], d['build_s'], d['graph_walk_queries_per_s']))
    print('walk kernel ms packed %.4f plain %.4f  equal %s  prefetch hits %.3f  expansions/query %.1f' % (r['kernel_ms'], r['plain_layout_kernel_ms'], r['packed_equals_plain_bit_exact'], r['prefetched_records_used'], r['expansions_per_query']))
    print('cycles per query by phase', r['cycles_per_query_by_phase'])
except Exception as e:
    print('ERR', e); print(open('gpurun_out/r05c5/bench_hnsw_5m.err').read()[-2000:])
PY
