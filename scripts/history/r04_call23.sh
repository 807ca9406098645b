#!/usr/bin/env bash
# round 4, call 23: the whole GPU suite with M = 32 on the byte-table kernel by default, then its timing through the library's own
# kernel choice (scan state: guarded first launch, then settled) at 10M and 1.25M rows
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r04c23; mkdir -p $OUT
timeout 215 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -6 $OUT/pytest_gpu.txt
timeout 40 python scripts/prof_scan.py --rows 10000000 --m 32 --dsub 4 --data lowrank --fused --valid --iters 8 --k 10 > $OUT/scan_10m_m32_default.txt 2>&1
grep -v "^/opt" $OUT/scan_10m_m32_default.txt | head -3 | cut -c1-300
