#!/usr/bin/env bash
# Round 6, call 35: the fused candidate generator -- tests, the re-rank tests, the re-rank leg with and without the table-wide seed.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c35; mkdir -p $OUT
timeout 900 python -m pytest tests/test_candidates_fused.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_fused.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_round4_gpu.py tests/test_fuzz_parity.py tests/test_sharded_gloo.py -x -q -m gpu -k "rerank or candidate or fuzz or sharded" 2>&1 | tail -5 | tee $OUT/pytest_rerank.txt
for v in new old new old; do
  E=""; [ $v = old ] && E="ANNLITE_NO_CAND_SEED=1"
  env $E timeout 400 python bench.py --legs rerank --cpu-queries 0 --steps 40 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['rerank']
print('$v: main %.0f q/s; rerank leg %9.0f q/s at recall@10 %.4f (global pool %.0f at %.4f)' % (d['value'], r['value'], r['recall_at_10'], r['global_pool']['value'], r['global_pool']['recall_at_10']))"
done | tee $OUT/rerank_leg_ab.txt
