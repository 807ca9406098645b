#!/usr/bin/env bash
# Round 6, call 65: the randomised cells run with the candidate lists of cells in parts; the bench's ivf leg with the re-rank variants.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c65; mkdir -p $OUT
timeout 150 python tests/fuzz_parity.py --cells --seconds 50 --seed 174 2>&1 | tail -6 | tee $OUT/fuzz_parity_cells_seed174_parts.txt
timeout 600 python bench.py --legs rerank,ivf --cpu-queries 0 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
for n in ('rerank', 'rerank16', 'rerank16_whole_cells', 'rerank16_rank1', 'rerank16_top16'): print(n, json.dumps(r['ivf'].get(n)))
print(json.dumps(r['summary']))" | tee $OUT/bench_ivf_leg.txt
