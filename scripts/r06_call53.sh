#!/usr/bin/env bash
# Round 6, call 53: cell selection's wave argmin by DPP; the bench's ivf leg on the shared stream pair.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c53; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ivf_byte_tiles.py tests/test_ivf.py tests/test_fuzz_parity.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -6 | tee $OUT/pytest_ivf.txt
timeout 600 python scripts/bench_ivf_bytes.py --probes 8,16,32 --reps 30 2>&1 | grep n_probe | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['n_probe'], r['bytes'], r['paths_bit_equal'], r['stages'])" | tee $OUT/ivf_dpp_select.txt
timeout 600 python bench.py --legs ivf --cpu-queries 0 2>/dev/null | tail -c 500 | tee $OUT/bench_ivf_leg_tail.txt
