#!/usr/bin/env bash
# Round 6, call 49: inner-product tables in the cells' preparation launch (COSINE / INNER_PRODUCT), randomised parity of the pruned search.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c49; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ivf_byte_tiles.py tests/test_ivf.py tests/test_fuzz_parity.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -15 | tee $OUT/pytest_ivf.txt
timeout 400 python tests/fuzz_parity.py --cells --seconds 90 --seed 71 2>&1 | tail -5 | tee $OUT/fuzz_cells_seed71.txt
timeout 600 python scripts/bench_ivf_bytes.py --probes 16 2>&1 | grep n_probe | tee $OUT/ivf_bytes_p16.txt
