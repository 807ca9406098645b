#!/usr/bin/env bash
# Round 6, call 42: the pruned search over cells on the byte-table kernel (annlite_ivf_search_topk) -- parity tests, then the 10M-row A/B
# against the u16 tile scan + re-score.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c42; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ivf_byte_tiles.py tests/test_ivf.py -x -q -m gpu 2>&1 | tail -25 | tee $OUT/pytest_ivf.txt
timeout 900 python scripts/bench_ivf_bytes.py --probes 8,16,32 2>&1 | tail -8 | tee $OUT/ivf_bytes_10m.txt
