#!/usr/bin/env bash
# Round 6, call 51: the cell tiles' fp32 tables per query ([b][Ks][M]) instead of TILED groups of four.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c51; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ivf_byte_tiles.py tests/test_ivf.py tests/test_fuzz_parity.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -6 | tee $OUT/pytest_ivf.txt
for i in 1 2; do timeout 600 python scripts/bench_ivf_bytes.py --probes 8,16,32 --reps 30 2>&1 | grep n_probe | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['n_probe'], r['bytes'], r['paths_bit_equal'], r['stages'])" | tee -a $OUT/ivf_per_query_tables.txt; done
