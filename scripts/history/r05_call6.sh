#!/usr/bin/env bash
# Round 5, GPU call 6: the whole GPU suite on the current tree; k = 50 whole-call time (vectorised seed selection); config 5.
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r05c6; mkdir -p $OUT
timeout 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -4 $OUT/pytest_gpu.txt
P="--rows 10000000 --data lowrank --fused --valid --iters 8"
timeout 90 python scripts/prof_scan.py $P --k 50 2>&1 | grep -v "^/opt" | head -3 > $OUT/scan_10m_k50_library_choice.txt
ANNLITE_SCAN_VARIANT=50 ANNLITE_DEBUG_COUNTERS=2 timeout 90 python scripts/prof_scan.py $P --k 50 2>&1 | grep -v "^/opt" | head -6 > $OUT/scan_10m_k50_timeline.txt
timeout 90 python scripts/prof_scan.py $P --k 10 2>&1 | grep -v "^/opt" | head -3 > $OUT/scan_10m_k10.txt
for f in $OUT/scan_10m_*.txt; do echo "== $f"; cut -c1-300 $f; done
timeout 600 python scripts/bench_hnsw.py --rows 5000000 --steps 5 > $OUT/bench_hnsw_5m.json 2>$OUT/bench_hnsw_5m.err; echo "bench_hnsw rc=$?"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/r05c6/bench_hnsw_5m.json') if l.startswith('{')][-1])
    r = d['roofline']
    print('c5: %.0f q/s recall %.3f build %.1f s; walk q/s %s' % (d['value'], d['recall_at_10'], d['build_s'], d['graph_walk_queries_per_s']))
    print('walk kernel ms packed %.4f plain %.4f  equal %s  prefetch hits %.3f  expansions/query %.1f' % (r['kernel_ms'], r['plain_layout_kernel_ms'], r['packed_equals_plain_bit_exact'], r['prefetched_records_used'], r['expansions_per_query']))
    print('cycles per query by phase', r['cycles_per_query_by_phase'])
except Exception as e:
    print('ERR', e); print(open('gpurun_out/r05c6/bench_hnsw_5m.err').read()[-2000:])
PY
