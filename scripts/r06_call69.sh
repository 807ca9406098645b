#!/usr/bin/env bash
# Round 6, call 69: the first bound from the WHOLE nearest cell while its parts are probed (seed_cells): tests, fuzz, the sweep.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c69; mkdir -p $OUT
timeout 300 python -m pytest tests/test_ivf_byte_tiles.py -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -40 | tee $OUT/pytest_ivf.txt
timeout 100 python tests/fuzz_parity.py --cells --seconds 30 --seed 175 2>&1 | tail -6 | tee $OUT/fuzz_parity_cells_seed175_seed_cells.txt
timeout 300 python scripts/sweep_ivf_rerank.py --configs 1:2x4p,1:2x4,2:2x4,1:4x4,2:4x4,1:2x8,4:2x4 2>&1 | grep "^{\|Error\|error" | tee $OUT/ivf_rerank_seed_whole_cell_sweep.txt
