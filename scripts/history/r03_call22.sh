#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c22; mkdir -p $OUT
export TMPDIR=/tmp
P="python scripts/prof_scan.py --data lowrank --fused --valid --rows 1250000 --iters 12"
for sk in 0 1 2 3 4; do
  echo "== SKIP=$sk"; ANNLITE_DEBUG_SKIP=$sk $P 2>/dev/null | grep "scan kernel ms" | cut -c1-120
  ANNLITE_DEBUG_COUNTERS=2 ANNLITE_DEBUG_SKIP=$sk $P 2>/dev/null | grep "timeline" | cut -c1-300
done
