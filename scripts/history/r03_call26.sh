#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
P="python scripts/prof_scan.py --data lowrank --fused --valid --rows 1250000 --iters 6"
for ns in 1 2 4 8 16; do
  echo "== slices $ns"; ANNLITE_SCAN_SLICES=$ns ANNLITE_DEBUG_COUNTERS=1 $P 2>/dev/null | grep "scan kernel ms\|byte-table kernel: wave" | cut -c1-260
done
echo "== slices 8, import every batch"; ANNLITE_Q8_TUNE="15,16,384,0" ANNLITE_DEBUG_COUNTERS=1 $P 2>/dev/null | grep "scan kernel ms\|byte-table kernel: wave" | cut -c1-260
echo "== slices 8, seed 8192"; ANNLITE_SEED_ROWS=8192 ANNLITE_DEBUG_COUNTERS=1 $P 2>/dev/null | grep "scan kernel ms\|byte-table kernel: wave" | cut -c1-260
