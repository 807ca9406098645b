#!/usr/bin/env bash
# Round 6, call 21: three-register lists for 128 < ef <= 192 (pair walk): tests, the ef sweep at 10M rows again.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c21; mkdir -p $OUT
timeout 900 python -m pytest tests/test_graph_pair.py tests/test_graph_packed.py tests/test_graph_gpu_build.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 600 python scripts/graph_build_probe.py --rows 10000000 --seeds 128 --ef 128,144,160,192,200 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_10m_ef.txt
