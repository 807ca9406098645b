#!/usr/bin/env bash
# Round 6, call 20: the insertion walks (ef_construction 200) with a 4096-entry visited table (four waves per CU) against 8192 (three);
# the graph tests on the library with 4096 entries up to ef = 192.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c20; mkdir -p $OUT
timeout 900 python -m pytest tests/test_graph_pair.py tests/test_graph_packed.py tests/test_graph_gpu_build.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.txt
for hb in default 12 default 12; do
  E=""; [ $hb != default ] && E="ANNLITE_GRAPH_HASH_BITS=$hb"
  echo "== insertion walks with the visited table: $hb"
  env $E timeout 600 python scripts/graph_build_probe.py --rows 5000000 --seeds 128 --ef 128 2>&1 | grep -v amdgpu.ids
done | tee $OUT/build_hash_bits.txt
