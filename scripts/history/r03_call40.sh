#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
P="python scripts/prof_scan.py --data lowrank --fused --valid --m 64 --batch 256 --rows 10000000 --iters 8"
echo "== M=64 normal"; $P 2>/dev/null | grep "scan kernel ms" | cut -c1-90
ANNLITE_DEBUG_COUNTERS=1 $P 2>/dev/null | grep "byte-table kernel: wave" | cut -c1-260
echo "== M=64 no candidate handling"; ANNLITE_DEBUG_SKIP=4 $P 2>/dev/null | grep "scan kernel ms" | cut -c1-90
P="python scripts/prof_scan.py --data lowrank --fused --valid --m 8 --dsub 16 --ks 512 --rows 10000000 --iters 8"
echo "== M=8 ks512 normal"; $P 2>/dev/null | grep "scan kernel ms" | cut -c1-90
ANNLITE_DEBUG_COUNTERS=1 $P 2>/dev/null | grep "byte-table kernel: wave" | cut -c1-260
echo "== M=8 no candidate handling"; ANNLITE_DEBUG_SKIP=4 $P 2>/dev/null | grep "scan kernel ms" | cut -c1-90
