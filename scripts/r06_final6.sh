#!/usr/bin/env bash
# Round 6, final evidence of the tree (fifth session): the nearest cells in parts for the re-rank on the cell tiles.
# The whole GPU suite, the default bench line.
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=gpurun_out/r06f6; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/ -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -25 | tee $OUT/pytest_gpu_suite.txt
timeout 900 python bench.py > $OUT/bench_10m_n1.json 2>$OUT/bench_10m_n1.err; tail -c 1500 $OUT/bench_10m_n1.json; echo
