#!/usr/bin/env bash
# Round 5, GPU call 16: slice-per-XCD map for M = 32 (HBM 5.7 GB per 10M-row launch with the tile-per-XCD map): time A/B + traffic, one box.
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r05c16; mkdir -p $OUT
P="--rows 10000000 --m 32 --dsub 4 --data lowrank --fused --valid --iters 8"
for map in 0 1 0 1; do
  ANNLITE_Q8_MAP=$map timeout 90 python scripts/prof_scan.py $P 2>&1 | grep -v "^/opt" | head -3 | tr '\n' ' ' > $OUT/m32_map${map}.txt; echo "m32 map $map: $(cut -c1-220 $OUT/m32_map${map}.txt)"
done
ANNLITE_Q8_MAP=1 bash scripts/gpu_pmc_traffic.sh r05_m32_map1 $P > $OUT/traffic_m32_map1.txt 2>&1; cat $OUT/traffic_m32_map1.txt
for rows in 2500000; do for map in 0 1; do
  ANNLITE_Q8_MAP=$map timeout 90 python scripts/prof_scan.py --rows $rows --m 32 --dsub 4 --data lowrank --fused --valid --iters 12 2>&1 | grep -v "^/opt" | head -3 | tr '\n' ' ' > $OUT/m32_${rows}_map${map}.txt; echo "m32 $rows map $map: $(cut -c1-220 $OUT/m32_${rows}_map${map}.txt)"
done; done
find gpurun_out -name "*.db" -delete 2>/dev/null; find gpurun_out -name "*kernel_trace.csv" -delete 2>/dev/null; find gpurun_out -name "*agent_info.csv" -delete 2>/dev/null
