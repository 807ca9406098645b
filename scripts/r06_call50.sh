#!/usr/bin/env bash
# Round 6, call 50: cell tiles -- table target x seed rows, 10M rows, 16 of 256 cells (same box).
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c50; mkdir -p $OUT
run() { echo "$*" | tee -a $OUT/ivf_target_seed.txt; env "$@" timeout 600 python scripts/bench_ivf_bytes.py --probes 16 --reps 30 2>&1 | grep n_probe | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['n_probe'], r['bytes'], r['stages'])" | tee -a $OUT/ivf_target_seed.txt; }
run ANNLITE_X=0
run ANNLITE_Q8_TARGET=64
run ANNLITE_Q8_TARGET=72
run ANNLITE_Q8_TARGET=80
run ANNLITE_SEED_ROWS=49152
run ANNLITE_SEED_ROWS=49152 ANNLITE_Q8_TARGET=72
run ANNLITE_X=0
