#!/usr/bin/env bash
# Round 6, call 45: what bounds the cell-tile scan's consumer -- no table gathers (wrong results, timing only), more seed rows.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c45; mkdir -p $OUT
run() { echo "$1" | tee -a $OUT/ivf_knobs.txt; env $1 ANNLITE_IVF_FIRST=0 timeout 600 python scripts/bench_ivf_bytes.py --probes 16 2>&1 | grep n_probe | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['n_probe'], r['bytes'], r['paths_bit_equal'], r['agreement_with_exhaustive_adc_topk'], r['stages'])" | tee -a $OUT/ivf_knobs.txt; }
run ANNLITE_X=0
run ANNLITE_DEBUG_SKIP=1
run ANNLITE_DEBUG_SKIP=3
run ANNLITE_SEED_ROWS=16384
run ANNLITE_SEED_ROWS=65536
run ANNLITE_SEED_ROWS=131072
run ANNLITE_Q8_TARGET=64
run ANNLITE_Q8_TARGET=112
