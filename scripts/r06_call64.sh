#!/usr/bin/env bash
# Round 6, call 64: the nearest cells in parts (IvfPQGpuIndex.rerank_split) -- the cell-tile tests, then pool shape against rate and recall.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c64; mkdir -p $OUT
timeout 300 python -m pytest tests/test_ivf_byte_tiles.py -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -40 | tee $OUT/pytest_ivf.txt
timeout 400 python scripts/sweep_ivf_rerank.py 2>&1 | grep "^{\|Error\|error" | tee $OUT/ivf_rerank_split_sweep.txt
