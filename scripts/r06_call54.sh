#!/usr/bin/env bash
# Round 6, call 54: the cell tiles as the candidate generator of the float re-rank (private lists): tests, the bench's ivf leg.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c54; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ivf_byte_tiles.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -12 | tee $OUT/pytest_ivf.txt
timeout 600 python bench.py --legs rerank,ivf --cpu-queries 0 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print(json.dumps(r['ivf'])); print(json.dumps(r['summary']))" | tee $OUT/bench_ivf_leg.txt
