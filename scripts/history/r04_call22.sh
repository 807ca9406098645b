#!/usr/bin/env bash
# round 4, call 22: first run of the M = 32 byte-table shape (adc_scan_q8_kernel<32,16,*,1,1>, ANNLITE_SCAN_VARIANT=50): its own test
# file, the older scan tests under the variant, then the 10M-row timing next to the u16 kernel's 7.41 ms
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r04c22; mkdir -p $OUT
timeout 150 python -m pytest tests/test_m32_byte_tables.py -x -q -m gpu > $OUT/pytest_m32.txt 2>&1; echo "m32 tests rc=$?"; tail -15 $OUT/pytest_m32.txt
ANNLITE_SCAN_VARIANT=50 timeout 100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_shapes or fused_search_entry" > $OUT/pytest_parity_v50.txt 2>&1; echo "parity v50 rc=$?"; tail -4 $OUT/pytest_parity_v50.txt
ANNLITE_SCAN_VARIANT=50 timeout 60 python scripts/prof_scan.py --rows 10000000 --m 32 --dsub 4 --data lowrank --fused --valid --iters 8 --k 10 > $OUT/scan_10m_m32_q8.txt 2>&1
grep -v "^/opt" $OUT/scan_10m_m32_q8.txt | head -3 | cut -c1-300
