#!/usr/bin/env bash
# Round 6, final evidence of the tree (fourth session, second pass): the candidate generator's bound_rank became an argument of the C entry.
# The cell-tile tests, the randomised cells run with the candidate lists' checks, the whole GPU suite, the default bench line.
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=gpurun_out/r06f5; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ivf_byte_tiles.py -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -30 | tee $OUT/pytest_ivf.txt
timeout 200 python tests/fuzz_parity.py --cells --seconds 60 --seed 172 2>&1 | tail -12 | tee $OUT/fuzz_parity_cells_seed172.txt
timeout 1200 python -m pytest tests/ -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -25 | tee $OUT/pytest_gpu_suite.txt
timeout 900 python bench.py > $OUT/bench_10m_n1.json 2>$OUT/bench_10m_n1.err; tail -c 1500 $OUT/bench_10m_n1.json; echo
