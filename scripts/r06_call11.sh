#!/usr/bin/env bash
# Round 6, call 11: the pair walk (two nodes per step of the packed graph walk) -- its tests, the C5 tests, and the C5 leg at 5M rows
# with the one-at-a-time walk timed beside it.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c11; mkdir -p $OUT
timeout 900 python -m pytest tests/test_graph_pair.py tests/test_graph_packed.py tests/test_k64_stress.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_graph.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hnsw or graph" --deselect "tests/test_gpu_parity.py::test_config5_hnsw_pq_oracle_side[5000000-0.85-0.88]" 2>&1 | tail -8 | tee $OUT/pytest_c5.txt
timeout 900 python scripts/bench_hnsw.py --rows 5000000 --steps 5 > $OUT/bench_hnsw_5m.json 2> $OUT/bench_hnsw_5m.err
tail -3 $OUT/bench_hnsw_5m.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06c11/bench_hnsw_5m.json') if l.startswith('{')][-1])
r = d['roofline']
print('c5: %.0f q/s recall %.4f build_s %.1f' % (d['value'], d.get('recall_at_10', -1), d.get('build_s', -1)))
print('walk kernel_ms %.4f (one at a time %.4f, plain %.4f) width %s overlap %s expansions/query %.1f rows/query %.1f prefetch used %.3f' % (
    r['kernel_ms'], r.get('one_at_a_time_kernel_ms') or -1, r['plain_layout_kernel_ms'], r.get('expand_width'),
    r.get('pair_candidates_shared_with_one_at_a_time'), r['expansions_per_query'], r['rows_evaluated_per_query'], r['prefetched_records_used']))
print('cycles/query', r['cycles_per_query_by_phase'])
for k, v in ((k, d[k]) for k in d if k.startswith('hnsw_') or k.startswith('exhaustive')):
    print(k, v)
PY
