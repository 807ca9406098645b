#!/usr/bin/env bash
# Round 6, call 68: the re-ranked pruned search at 8 and 32 probed cells (pool shape against rate and recall).
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c68; mkdir -p $OUT
for P in 8 32; do
  echo "n_probe $P" | tee -a $OUT/ivf_rerank_split_sweep_probes.txt
  timeout 200 python scripts/sweep_ivf_rerank.py --probe $P --configs 1:0x1,1:2x4,1:4x4,2:2x4,1:2x8 2>&1 | grep "^{\|Error\|error" | tee -a $OUT/ivf_rerank_split_sweep_probes.txt
done
