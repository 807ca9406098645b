#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
export TMPDIR=/tmp
P="python scripts/prof_scan.py --data lowrank --fused --valid --rows 1250000 --iters 16"
for v in base bu4 bu8; do
  [ $v = base ] && unset ANNLITE_HIP_LIB || export ANNLITE_HIP_LIB=$ROOT/build_exp/lib_$v.so
  echo "== $v"; $P 2>/dev/null | grep "scan kernel ms" | cut -c1-90
  ANNLITE_DEBUG_COUNTERS=2 $P 2>/dev/null | grep "timeline" | cut -c60-330
done
