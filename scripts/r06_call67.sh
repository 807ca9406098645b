#!/usr/bin/env bash
# Round 6, call 67: the pool's keywords through the facade (AnnLite(..., ivf_prune=True, rerank=True, rerank_split=...)).
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c67; mkdir -p $OUT
timeout 300 python -m pytest tests/test_ivf_byte_tiles.py -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -40 | tee $OUT/pytest_ivf.txt
