#!/usr/bin/env bash
# Round 5, GPU call 23: where the M = 32 rule (8 slices + slice-per-XCD) stops paying: 5M and 2.5M rows, interleaved on one box
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; mkdir -p gpurun_out/r05c23
for rows in 5000000 2500000; do
  P="--rows $rows --m 32 --dsub 4 --data lowrank --fused --valid --iters 12"
  for rep in 1 2; do
    echo "m32 $rows planned #$rep: $(timeout 90 python scripts/prof_scan.py $P 2>&1 | grep -v '^/opt' | head -3 | tr '\n' ' ' | cut -c1-200)"
    echo "m32 $rows 8 slices + map #$rep: $(ANNLITE_SCAN_SLICES=8 ANNLITE_Q8_MAP=1 timeout 90 python scripts/prof_scan.py $P 2>&1 | grep -v '^/opt' | head -3 | tr '\n' ' ' | cut -c1-200)"
  done
done
