#!/usr/bin/env bash
# Round 6, call 13: GPU graph build by phase, seeds of the walk, the graph tests with the GPU build as the default.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c13; mkdir -p $OUT
PROBE_PHASES=1 timeout 600 python scripts/graph_build_probe.py --rows 5000000 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_5m.txt
timeout 600 python scripts/graph_build_probe.py --rows 5000000 --build-seeds 128 --seeds 128 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_5m_build_seeds128.txt
timeout 600 python scripts/graph_build_probe.py --rows 5000000 --build-seeds 128 --seeds 128 --batch 65536 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_5m_batch64k.txt
timeout 1500 python -m pytest tests/test_graph_gpu_build.py tests/test_graph_pair.py tests/test_graph_packed.py tests/test_gpu_parity.py -x -q -m gpu -k "graph or hnsw or config5" 2>&1 | tail -8 | tee $OUT/pytest.txt
