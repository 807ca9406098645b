#!/usr/bin/env bash
# round 4, call 25: the M = 32 shape's forced epochs / rebuilds and the guarded give-up path (tests added after the full-suite run)
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r04c25; mkdir -p $OUT
timeout 40 python -m pytest tests/test_m32_byte_tables.py -x -q -m gpu -k "epochs or give_up or candidate" > $OUT/pytest_m32_more.txt 2>&1; echo "rc=$?"; tail -12 $OUT/pytest_m32_more.txt
