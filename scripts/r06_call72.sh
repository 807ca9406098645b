#!/usr/bin/env bash
# Round 6, call 72: the plan kernel keeps 24 pairs per thread in registers (22 entries per query with the nearest two cells in four parts).
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=gpurun_out/r06c72; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_ivf_byte_tiles.py tests/test_ivf.py -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -8 | tee $OUT/pytest_ivf.txt
timeout 60 python tests/fuzz_parity.py --cells --seconds 20 --seed 176 2>&1 | tail -3 | tee $OUT/fuzz_parity_cells_seed176.txt
timeout 100 python bench.py --legs rerank,ivf --cpu-queries 0 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
for n in ('rerank16', 'rerank16_rank2'): print(n, json.dumps(r['ivf'].get(n)))
print(json.dumps(r['summary']))" | tee $OUT/bench_ivf_leg.txt
