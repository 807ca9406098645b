#!/usr/bin/env bash
# Round 5, GPU call 11: two-level seed selection for k > 16 + import every 8th batch: tests, k = 50 / 64 / 20 timings, preparation timeline.
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r05c11; mkdir -p $OUT
timeout 600 python -m pytest tests/test_k64_byte_tables.py tests/test_round4_gpu.py -x -q -m gpu > $OUT/pytest_k64_r4.txt 2>&1; echo "k64 + round4 rc=$?"; tail -3 $OUT/pytest_k64_r4.txt
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_shapes or ties or topk" > $OUT/pytest_parity_k.txt 2>&1; echo "parity rc=$?"; tail -3 $OUT/pytest_parity_k.txt
P="--rows 10000000 --data lowrank --fused --valid --iters 8"
for k in 50 64 20; do
  ANNLITE_SCAN_VARIANT=50 timeout 90 python scripts/prof_scan.py $P --k $k 2>&1 | grep -v "^/opt" | head -3 | tr '\n' ' ' > $OUT/scan_10m_k${k}_q8lk64.txt; echo "k=$k: $(cut -c1-200 $OUT/scan_10m_k${k}_q8lk64.txt)"
done
ANNLITE_SCAN_VARIANT=50 ANNLITE_Q8_TARGET=112 timeout 90 python scripts/prof_scan.py $P --k 50 2>&1 | grep -v "^/opt" | head -3 | tr '\n' ' ' > $OUT/scan_10m_k50_target112.txt; echo "k=50 T112: $(cut -c1-200 $OUT/scan_10m_k50_target112.txt)"
ANNLITE_SCAN_VARIANT=31 timeout 90 python scripts/prof_scan.py $P --k 50 2>&1 | grep -v "^/opt" | head -3 | tr '\n' ' ' > $OUT/scan_10m_k50_u16.txt; echo "k=50 u16: $(cut -c1-200 $OUT/scan_10m_k50_u16.txt)"
ANNLITE_SCAN_VARIANT=50 ANNLITE_DEBUG_COUNTERS=2 timeout 90 python scripts/prof_scan.py $P --k 50 2>&1 | grep "preparation launch" | cut -c1-330
ANNLITE_DEBUG_COUNTERS=2 timeout 90 python scripts/prof_scan.py $P --k 10 2>&1 | grep "preparation launch" | cut -c1-330
