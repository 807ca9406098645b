#!/usr/bin/env bash
# Round 6, call 44: the nearest cells' tiles first (ANNLITE_IVF_FIRST = 0 / 1 / 2 / 4) and the import interval, 10M rows, 16 of 256 cells.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c44; mkdir -p $OUT
timeout 600 python -m pytest tests/test_ivf_byte_tiles.py tests/test_ivf.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest_ivf.txt
for f in 0 1 2 4; do
  echo "ANNLITE_IVF_FIRST=$f" | tee -a $OUT/ivf_first.txt
  ANNLITE_IVF_FIRST=$f timeout 600 python scripts/bench_ivf_bytes.py --probes 8,16,32 2>&1 | grep n_probe | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['n_probe'], r['bytes'], r['paths_bit_equal'], r['stages'])" | tee -a $OUT/ivf_first.txt
done
for t in 255,16,384,0 255,16,384,1 255,16,384,7; do
  echo "ANNLITE_Q8_TUNE=$t (IVF_FIRST default)" | tee -a $OUT/ivf_first.txt
  ANNLITE_Q8_TUNE=$t timeout 600 python scripts/bench_ivf_bytes.py --probes 16 2>&1 | grep n_probe | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['n_probe'], r['bytes'], r['paths_bit_equal'], r['stages'])" | tee -a $OUT/ivf_first.txt
done
ANNLITE_DEBUG_COUNTERS=1 timeout 600 python scripts/prof_ivf_bytes.py --probe 16 2>&1 | tail -8 | tee $OUT/ivf_tiles_p16.txt
