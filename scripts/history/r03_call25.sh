#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
P="python scripts/prof_scan.py --data random --fused --valid --rows 10000000 --iters 8"
echo "== u16 forced"; ANNLITE_SCAN_VARIANT=31 $P 2>/dev/null | grep "scan kernel ms\|kernel choice" | cut -c1-150
echo "== byte forced (unguarded)"; ANNLITE_SCAN_VARIANT=50 timeout 120 $P 2>/dev/null | grep "scan kernel ms\|kernel choice" | cut -c1-150
echo "== library (state)"; $P 2>/dev/null | grep "scan kernel ms\|kernel choice\|ms per call" | cut -c1-200
