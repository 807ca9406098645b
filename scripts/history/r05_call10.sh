#!/usr/bin/env bash
# Round 5, GPU call 10: knobs of the k = 50 scan (import frequency, published positions, slices) and of the graph walk (visited-table size), one box.
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r05c10; mkdir -p $OUT
P="--rows 10000000 --data lowrank --fused --valid --iters 8 --k 50"
run() { tag=$1; shift; env "$@" ANNLITE_SCAN_VARIANT=50 timeout 90 python scripts/prof_scan.py $P 2>&1 | grep -v "^/opt" | head -3 | tr '\n' ' ' > $OUT/k50_$tag.txt; echo "k50 $tag: $(cut -c1-200 $OUT/k50_$tag.txt)"; }
run base X=1
run import1 ANNLITE_Q8_TUNE=15,16,384,1
run import7 ANNLITE_Q8_TUNE=15,16,384,7
run import15 ANNLITE_Q8_TUNE=15,16,384,15
run pos_tight ANNLITE_Q8_POS=0.5,0.8,1.1,1.6
run pos_wide ANNLITE_Q8_POS=0.8,1.1,1.5,2.4
run slices16 ANNLITE_SCAN_SLICES=16
run target112 ANNLITE_Q8_TARGET=112
run target80 ANNLITE_Q8_TARGET=80
timeout 600 python scripts/prof_graph_walk.py --build /tmp/g5m --rows 5000000 > $OUT/graph_build.log 2>&1
for hb in 12 13 11; do
  ANNLITE_GRAPH_HASH_BITS=$hb timeout 120 python scripts/prof_graph_walk.py --walk /tmp/g5m --rows 5000000 --iters 8 2>&1 | grep "graph walk" | cut -c1-330 > $OUT/walk_hash$hb.txt; echo "walk hash_bits $hb: $(cat $OUT/walk_hash$hb.txt)"
done
