#!/usr/bin/env bash
# Round 6, call 60: the two tests that failed with the candidate generator's bound at rank k -- full output.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c60; mkdir -p $OUT
timeout 300 python -m pytest tests/test_ivf_byte_tiles.py -q -m gpu -k float_rerank 2>&1 | grep -v "^  File\|^Extension" | tail -40 | tee $OUT/pytest_ivf.txt
timeout 200 python tests/fuzz_parity.py --cells --seconds 40 --seed 171 2>&1 | tail -30 | tee $OUT/fuzz_cells.txt
ANNLITE_IVF_CAND_RANK=4 timeout 200 python tests/fuzz_parity.py --cells --seconds 40 --seed 171 2>&1 | tail -30 | tee $OUT/fuzz_cells_rank4.txt
