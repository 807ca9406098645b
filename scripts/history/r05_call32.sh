#!/usr/bin/env bash
# Round 5, GPU call 32: seed rows in runs of 2^c blocks (ANNLITE_SEED_CHUNK_LOG) -- the new tests, then run length A/B on the
# bench's tables (1.25M / 1M / 10M rows, two streams) against the first-rows seed, and the table in cluster order again
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; OUT=gpurun_out/r05c32; mkdir -p $OUT
timeout 600 python -m pytest tests/test_seed_rows_spread.py "tests/test_gpu_parity.py::test_library_picks_the_scan_kernel" tests/test_round4_gpu.py -m gpu -x -q > $OUT/pytest_subset.txt 2>&1
echo "subset rc=$?"; tail -4 $OUT/pytest_subset.txt; cat gpurun_out/seed_rows_spread_test.txt
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']
    print('ms/step %.4f  q/s %.0f  kernel_ms %.4f  sha %s' % (d['ms_per_step'], d['value'], r['kernel_ms'], d['result_sha256'][:10]))
except Exception as e: print('ERR', e)
PY
}
C="--legs none --cpu-queries 0 --recall-queries 0 --no-rerank --streams 2 --steps 200 --warmup 20"
for rows in 1250000 1000000 10000000; do
  for V in c3 c0 c5 contig contig_e15 c3; do
    case $V in
      c3) E="";; c0) E="ANNLITE_SEED_CHUNK_LOG=0";; c5) E="ANNLITE_SEED_CHUNK_LOG=5";; contig) E="ANNLITE_SEED_CONTIGUOUS=1";;
      contig_e15) E="ANNLITE_SEED_CONTIGUOUS=1 ANNLITE_Q8_TUNE=15,16,384,3";;
    esac
    env $E timeout 200 python bench.py --rows $rows $C > $OUT/t_${rows}_$V.json 2>/dev/null; echo "rows $rows $V: $(line $OUT/t_${rows}_$V.json)"
  done
done
for rows in 1250000 10000000; do
  for V in c3 c0 c5 contig; do
    case $V in c3) E="";; c0) E="ANNLITE_SEED_CHUNK_LOG=0";; c5) E="ANNLITE_SEED_CHUNK_LOG=5";; contig) E="ANNLITE_SEED_CONTIGUOUS=1";; esac
    echo "sorted rows $rows $V: $(env $E timeout 200 python scripts/prof_scan.py --rows $rows --fused --data lowrank --order sorted --iters 12 2>&1 | grep -E 'whole call|scan kernel ms' | tr '\n' ' ')"
  done
done
