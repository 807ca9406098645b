#!/usr/bin/env bash
# Round 6, call 14: the fused re-rank (annlite_rerank_topk) -- its test, the graph / re-rank tests, C5 at 5M rows, the re-rank leg.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c14; mkdir -p $OUT
timeout 900 python -m pytest tests/test_rerank_topk.py tests/test_graph_gpu_build.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest_rerank.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_round4_gpu.py -x -q -m gpu -k "rerank or hnsw or config5 or graph or facade" 2>&1 | tail -8 | tee $OUT/pytest_parity.txt
timeout 900 python scripts/bench_hnsw.py --rows 5000000 --steps 20 > $OUT/bench_hnsw_5m.json 2> $OUT/bench_hnsw_5m.err
tail -3 $OUT/bench_hnsw_5m.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06c14/bench_hnsw_5m.json') if l.startswith('{')][-1])
r = d['roofline']
print(' c5: %.0f q/s recall %.4f built on %s: build_s %.1f (gpu %s host %s)' % (d['value'], d['recall_at_10'], d['graph_built_on'], d['build_s'], d['gpu_build_s'], d['host_build_s']))
print(' walk kernel_ms %.4f (one at a time %s) expansions/query %.1f rows/query %.1f' % (r['kernel_ms'], r.get('one_at_a_time_kernel_ms'), r['expansions_per_query'], r['rows_evaluated_per_query']))
for k in d:
    if k.startswith('hnsw_') or k.startswith('exhaustive'):
        print(' ', k, d[k])
PY
timeout 600 python bench.py --legs rerank --cpu-queries 0 --steps 40 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['rerank']
print('main %.0f q/s; rerank leg: %9.0f q/s at recall@10 %.4f' % (d['value'], r['value'], r['recall_at_10']))" | tee $OUT/rerank_leg.txt
