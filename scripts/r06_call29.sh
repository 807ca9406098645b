#!/usr/bin/env bash
# Round 6, call 29: the GPU suite and the default bench line + rocprofv3 kernel stats on the round's final tree.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c29; mkdir -p $OUT
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest_gpu_suite.txt
bash scripts/r06_profiles.sh bench stats 2>&1 | grep -v "at::native\|rocprim\|rocclr\|Cijk" | tail -12
