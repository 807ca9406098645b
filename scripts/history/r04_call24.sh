#!/usr/bin/env bash
# round 4, call 24: M = 32 at 10M rows on ONE box: the byte-table kernel pinned (variant 50), the library's own choice, the u16 kernel (variant 31)
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r04c24; mkdir -p $OUT
A="--rows 10000000 --m 32 --dsub 4 --data lowrank --fused --valid --iters 8 --k 10"
ANNLITE_SCAN_VARIANT=50 timeout 30 python scripts/prof_scan.py $A 2>&1 | grep -v "^/opt" | head -2 | cut -c1-200 | tee $OUT/v50.txt
timeout 30 python scripts/prof_scan.py $A 2>&1 | grep -v "^/opt" | head -2 | cut -c1-200 | tee $OUT/default.txt
ANNLITE_SCAN_VARIANT=31 timeout 30 python scripts/prof_scan.py $A 2>&1 | grep -v "^/opt" | head -2 | cut -c1-200 | tee $OUT/v31.txt
