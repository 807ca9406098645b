#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
export TMPDIR=/tmp
for rows in 1250000 10000000; do
P="python scripts/prof_scan.py --data lowrank --fused --valid --rows $rows --iters 12"
for v in base exp5; do
  [ $v = base ] && unset ANNLITE_HIP_LIB || export ANNLITE_HIP_LIB=$ROOT/build_exp/lib_$v.so
  echo "== $rows $v"; $P 2>/dev/null | grep "scan kernel ms" | cut -c1-90
  ANNLITE_DEBUG_COUNTERS=2 $P 2>/dev/null | grep "timeline" | cut -c60-330
  ANNLITE_DEBUG_COUNTERS=1 $P 2>/dev/null | grep "byte-table kernel: wave" | cut -c1-230
done; done
