#!/usr/bin/env bash
# Round 6, call 46: cell tiles drawn from a counter (longest first), the plan inside the preparation launch, 65536 seed rows.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c46; mkdir -p $OUT
timeout 600 python -m pytest tests/test_ivf_byte_tiles.py tests/test_ivf.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest_ivf.txt
run() { echo "$1" | tee -a $OUT/ivf_knobs.txt; env $1 timeout 600 python scripts/bench_ivf_bytes.py --probes ${2:-16} 2>&1 | grep n_probe | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['n_probe'], r['bytes'], r['paths_bit_equal'], r['stages'])" | tee -a $OUT/ivf_knobs.txt; }
run ANNLITE_X=0 8,16,32
run ANNLITE_IVF_STATIC_TILES=1
run ANNLITE_IVF_PLAN_APART=1
run ANNLITE_SEED_ROWS=32768
run ANNLITE_X=0
