#!/usr/bin/env bash
# round 4, call 20: how the byte-table kernel's consumer load grows with k (10 -> 16, its largest list) at 10M rows, and what the u16
# kernel takes at k = 50 on the same table -- the two numbers behind DESIGN section 10 item 2 (byte tables for 16 < k <= 64)
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r04c20; mkdir -p $OUT
ANNLITE_DEBUG_COUNTERS=1 timeout 100 python scripts/prof_scan.py --rows 10000000 --data lowrank --fused --valid --iters 8 --k 16 > $OUT/scan_10m_k16_debug_counters.txt 2>&1
grep -v "^/opt" $OUT/scan_10m_k16_debug_counters.txt | head -4 | cut -c1-400
timeout 60 python scripts/prof_scan.py --rows 10000000 --data lowrank --fused --valid --iters 6 --k 50 > $OUT/scan_10m_k50_u16.txt 2>&1
grep -v "^/opt" $OUT/scan_10m_k50_u16.txt | head -3 | cut -c1-300
