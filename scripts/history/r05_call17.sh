#!/usr/bin/env bash
# Round 5, GPU call 17: M = 32 with 8 row slices (two work items per workgroup) so that the slice-per-XCD map applies: time, one box, interleaved.
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r05c17; rm -rf gpurun_out/*; mkdir -p $OUT
P="--rows 10000000 --m 32 --dsub 4 --data lowrank --fused --valid --iters 10"
for rep in 1 2; do
  timeout 90 python scripts/prof_scan.py $P 2>&1 | grep -v "^/opt" | head -3 | tr '\n' ' ' > $OUT/m32_default_$rep.txt; echo "m32 4 slices (default) #$rep: $(cut -c1-200 $OUT/m32_default_$rep.txt)"
  ANNLITE_SCAN_SLICES=8 ANNLITE_Q8_MAP=0 timeout 90 python scripts/prof_scan.py $P 2>&1 | grep -v "^/opt" | head -3 | tr '\n' ' ' > $OUT/m32_s8_map0_$rep.txt; echo "m32 8 slices map 0 #$rep: $(cut -c1-200 $OUT/m32_s8_map0_$rep.txt)"
  ANNLITE_SCAN_SLICES=8 ANNLITE_Q8_MAP=1 timeout 90 python scripts/prof_scan.py $P 2>&1 | grep -v "^/opt" | head -3 | tr '\n' ' ' > $OUT/m32_s8_map1_$rep.txt; echo "m32 8 slices map 1 #$rep: $(cut -c1-200 $OUT/m32_s8_map1_$rep.txt)"
done
