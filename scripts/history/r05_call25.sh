#!/usr/bin/env bash
# Round 5, GPU call 25: seed rows for k = 50 (the seed bound's rank is k N / S: 32768 rows were tuned at k = 10), one box, interleaved
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; mkdir -p gpurun_out/r05c25
P="--rows 10000000 --data lowrank --fused --valid --iters 10 --k 50"
for rep in 1 2; do for S in 32768 65536 131072 262144; do
  echo "k50 seed rows $S #$rep: $(ANNLITE_SEED_ROWS=$S ANNLITE_SCAN_VARIANT=50 timeout 90 python scripts/prof_scan.py $P 2>&1 | grep -v '^/opt' | head -3 | tr '\n' ' ' | cut -c1-200)"
done; done
echo "k50 seed rows 131072 timeline: $(ANNLITE_SEED_ROWS=131072 ANNLITE_SCAN_VARIANT=50 ANNLITE_DEBUG_COUNTERS=1 timeout 90 python scripts/prof_scan.py $P 2>&1 | grep 'byte-table kernel:' | cut -c1-330)"
