#!/usr/bin/env bash
# Round 6, call 26: snapshots across the two graph builds; bench.py's graph leg as a test.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c26; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_graph_gpu_build.py tests/test_bench_two_ranks.py tests/test_gpu_parity.py -x -q -m gpu -k "snapshot or graph_leg or facade or dump or reopen or graph" 2>&1 | tail -6 | tee $OUT/pytest.txt
