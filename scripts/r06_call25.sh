#!/usr/bin/env bash
# Round 6, call 25: candidates of a step inserted one at a time when they are few (0 = always the merge, 2, 3 = the library, 6), same box.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c25; mkdir -p $OUT
timeout 600 python -m pytest tests/test_graph_pair.py tests/test_graph_packed.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do for v in few0 few2 lib few6; do
  E=""; [ $v != lib ] && E="ANNLITE_HIP_LIB=$PWD/build_exp/lib_$v.so"
  env $E timeout 300 python scripts/bench_hnsw.py --rows 5000000 --steps 20 --build gpu 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('%-5s c5 %.0f q/s recall %.4f walk kernel_ms %.4f (one at a time %.4f) cycles %s' % ('$v', d['value'], d['recall_at_10'], r['kernel_ms'], r['one_at_a_time_kernel_ms'], {k: int(v) for k, v in r['cycles_per_query_by_phase'].items()}))"
done; done 2>&1 | tee $OUT/few_inserts_ab.txt
