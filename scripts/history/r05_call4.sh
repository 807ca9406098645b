#!/usr/bin/env bash
# Round 5, GPU call 4: 16 < k <= 64 on the byte-table kernel (64-key lists): tests, k = 50 at 10M rows against the u16 tables;
# the graph walk with the bucketed visited table + phase cycles.
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r05c4; mkdir -p $OUT
timeout 600 python -m pytest tests/test_k64_byte_tables.py -x -q > $OUT/pytest_k64.txt 2>&1; echo "k64 rc=$?"; tail -5 $OUT/pytest_k64.txt
P="--rows 10000000 --data lowrank --fused --valid --iters 8"
ANNLITE_SCAN_VARIANT=50 timeout 90 python scripts/prof_scan.py $P --k 50 2>&1 | grep -v "^/opt" | head -3 > $OUT/scan_10m_k50_q8lk64.txt
ANNLITE_SCAN_VARIANT=31 timeout 90 python scripts/prof_scan.py $P --k 50 2>&1 | grep -v "^/opt" | head -3 > $OUT/scan_10m_k50_u16.txt
ANNLITE_SCAN_VARIANT=50 ANNLITE_DEBUG_COUNTERS=1 timeout 90 python scripts/prof_scan.py $P --k 50 2>&1 | grep -v "^/opt" | head -12 > $OUT/scan_10m_k50_q8lk64_counters.txt
ANNLITE_SCAN_VARIANT=50 timeout 90 python scripts/prof_scan.py $P --k 64 2>&1 | grep -v "^/opt" | head -3 > $OUT/scan_10m_k64_q8lk64.txt
ANNLITE_SCAN_VARIANT=50 timeout 90 python scripts/prof_scan.py $P --k 20 2>&1 | grep -v "^/opt" | head -3 > $OUT/scan_10m_k20_q8lk64.txt
timeout 90 python scripts/prof_scan.py $P --k 50 2>&1 | grep -v "^/opt" | head -3 > $OUT/scan_10m_k50_library_choice.txt
for f in $OUT/scan_10m_*.txt; do echo "== $f"; cut -c1-260 $f; done
timeout 300 python -m pytest tests/test_graph_packed.py -x -q > $OUT/pytest_graph_packed.txt 2>&1; echo "graph_packed rc=$?"; tail -3 $OUT/pytest_graph_packed.txt
timeout 600 python scripts/bench_hnsw.py --rows 5000000 --steps 5 > $OUT/bench_hnsw_5m.json 2>$OUT/bench_hnsw_5m.err; echo "bench_hnsw rc=$?"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/r05c4/bench_hnsw_5m.json') if l.startswith('{')][-1])
    r = d['roofline']
    print('c5: %.0f q/s recall %.3f build %.1f s; walk q/s %s' % (d['value'], d['recall_at_10'], d['build_s'], d['graph_walk_queries_per_s']))
    print('walk kernel ms packed %.4f plain %.4f  equal %s  prefetch hits %.3f  expansions/query %.1f' % (r['kernel_ms'], r['plain_layout_kernel_ms'], r['packed_equals_plain_bit_exact'], r['prefetched_records_used'], r['expansions_per_query']))
    print('cycles per query by phase', r['cycles_per_query_by_phase'])
except Exception as e:
    print('ERR', e); print(open('gpurun_out/r05c4/bench_hnsw_5m.err').read()[-2000:])
PY
