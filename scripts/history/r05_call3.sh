#!/usr/bin/env bash
# Round 5, GPU call 3: the graph walk with merge insertion (tests: packed / plain / one-at-a-time agree) and config 5 at 5M rows.
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r05c3; mkdir -p $OUT
timeout 400 python -m pytest tests/test_graph_packed.py -x -q > $OUT/pytest_graph_packed.txt 2>&1; echo "graph_packed rc=$?"; tail -3 $OUT/pytest_graph_packed.txt
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5 or hnsw or graph" > $OUT/pytest_graph_cases.txt 2>&1; echo "graph cases rc=$?"; tail -3 $OUT/pytest_graph_cases.txt
timeout 600 python scripts/bench_hnsw.py --rows 5000000 --steps 5 > $OUT/bench_hnsw_5m.json 2>$OUT/bench_hnsw_5m.err; echo "bench_hnsw rc=$?"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/r05c3/bench_hnsw_5m.json') if l.startswith('{')][-1])
    r = d['roofline']
    print('c5: %.0f q/s recall %.3f build %.1f s; walk q/s %s' % (d['value'], d['recall_at_10'], d['build_s'], d['graph_walk_queries_per_s']))
    print('walk kernel ms packed %.4f plain %.4f  equal %s  prefetch hits %.3f  expansions/query %.1f' % (r['kernel_ms'], r['plain_layout_kernel_ms'], r['packed_equals_plain_bit_exact'], r['prefetched_records_used'], r['expansions_per_query']))
except Exception as e:
    print('ERR', e); print(open('gpurun_out/r05c3/bench_hnsw_5m.err').read()[-2000:])
PY
