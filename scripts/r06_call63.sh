#!/usr/bin/env bash
# Round 6, call 63: the re-rank's pool = exactly the ADC top-16 (rerank_bound_rank 0: annlite_ivf_search_topk's ids -> annlite_rerank_topk)
# beside the private lists: the cell-tile tests and the bench's ivf leg.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c63; mkdir -p $OUT
timeout 300 python -m pytest tests/test_ivf_byte_tiles.py -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -30 | tee $OUT/pytest_ivf.txt
timeout 600 python bench.py --legs rerank,ivf --cpu-queries 0 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
for n in ('rerank', 'rerank16', 'rerank16_top16'): print(n, json.dumps(r['ivf'].get(n)))
print(json.dumps(r['summary']))" | tee $OUT/bench_ivf_leg.txt
