#!/usr/bin/env bash
# round 4, call 21: the u16 kernel at M = 32 (10M rows x 1024 queries, k = 10): the baseline a byte-table M = 32 shape would have to beat
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r04c21; mkdir -p $OUT
timeout 100 python scripts/prof_scan.py --rows 10000000 --m 32 --dsub 4 --data lowrank --fused --valid --iters 6 --k 10 > $OUT/scan_10m_m32_u16.txt 2>&1
grep -v "^/opt" $OUT/scan_10m_m32_u16.txt | head -3 | cut -c1-300
