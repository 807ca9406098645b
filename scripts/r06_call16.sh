#!/usr/bin/env bash
# Round 6, call 16: the whole GPU suite on the tree with the pair walk, the GPU graph build (default) and the fused re-rank; the graph
# walk's HBM traffic passes.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c16; mkdir -p $OUT
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_gpu_suite.txt
bash scripts/r06_profiles.sh graph 2>&1 | tail -20
cp gpurun_out/r06p/graph_walk_5m_pmc_summary.txt $OUT/ 2>/dev/null
