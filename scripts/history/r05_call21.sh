#!/usr/bin/env bash
# Round 5, GPU call 21: slice-per-XCD map for the k = 50 launch (3.1 GB of HBM traffic per launch), interleaved on one box
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; mkdir -p gpurun_out/r05c21
P="--rows 10000000 --data lowrank --fused --valid --iters 10 --k 50"
for rep in 1 2; do for map in 0 1; do
  echo "k50 map $map #$rep: $(ANNLITE_Q8_MAP=$map ANNLITE_SCAN_VARIANT=50 timeout 90 python scripts/prof_scan.py $P 2>&1 | grep -v '^/opt' | head -3 | tr '\n' ' ' | cut -c1-200)"
done; done
