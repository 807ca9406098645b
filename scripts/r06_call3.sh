#!/usr/bin/env bash
# Round 6, call 3: 64-key lists for M = 8 / 32 (tests + numbers at 10M rows), the opt-in MFMA seed's evidence (rocprofv3 kernel
# trace: the nomination launch's own duration), the host cost of a batch with the exchange on (cProfile), M = 128 / M = 64 numbers.
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=gpurun_out/r06c3; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_k64_m8_m32.py tests/test_seed_mfma.py -x -q > $OUT/pytest_new.txt 2>&1; echo "new tests rc=$?"; tail -12 $OUT/pytest_new.txt
timeout 900 python -m pytest tests -q -m gpu --maxfail=8 --deselect tests/test_seed_mfma.py --deselect tests/test_k64_m8_m32.py > $OUT/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -6 $OUT/pytest_gpu.txt
# k = 50 at 10M rows x 1024 queries: byte tables with 64-key lists (the library's choice) against the u16 tables
P="--rows 10000000 --data lowrank --fused --valid --iters 8 --k 50"
for shape in "--m 32 --dsub 4" "--m 8 --dsub 16" "--m 8 --dsub 16 --ks 512"; do
  tag=$(echo $shape | tr -d ' -')
  timeout 200 python scripts/prof_scan.py $P $shape 2>&1 | grep -v "^/opt" | tail -3 > $OUT/scan_10m_k50_${tag}_library.txt
  ANNLITE_SCAN_VARIANT=31 timeout 200 python scripts/prof_scan.py $P $shape 2>&1 | grep -v "^/opt" | tail -3 > $OUT/scan_10m_k50_${tag}_u16.txt
  echo "== $shape"; cat $OUT/scan_10m_k50_${tag}_library.txt $OUT/scan_10m_k50_${tag}_u16.txt
done
# the example's M = 128 / dsub 1 (generic kernel) and M = 64 / dsub 2 at 1M rows x 256 queries
for shape in "--m 128 --dsub 1" "--m 64 --dsub 2"; do
  tag=$(echo $shape | tr -d ' -')
  timeout 200 python scripts/prof_scan.py --rows 1000000 --batch 256 --data lowrank --fused --valid --iters 6 $shape 2>&1 | grep -v "^/opt" | tail -3 > $OUT/scan_1m_${tag}.txt
  echo "== $shape"; cat $OUT/scan_1m_${tag}.txt
done
# the MFMA-nominated seed (opt-in): kernel durations under rocprofv3
for mode in mfma exact; do
  E=""; [ $mode = mfma ] && E="ANNLITE_MFMA_SEED=1"
  env $E rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace_$mode -- python scripts/prof_scan.py --rows 1250000 --data lowrank --fused --valid --iters 30 > $OUT/trace_$mode.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob('$OUT/trace_$mode/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'seed' in r['Name'] or 'adc_scan' in r['Name']: print('$mode %-70s calls=%-4s avg_us=%8.1f min_us=%8.1f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done 2>&1 | tee $OUT/mfma_seed_kernel_stats.txt
# host cost of a batch with / without the exchange (tiny table: the host shows), cProfile of 200 batches
timeout 300 python scripts/host_overhead.py > $OUT/host_overhead.txt 2>&1; tail -40 $OUT/host_overhead.txt
find gpurun_out -name '*.db' -delete 2>/dev/null; find gpurun_out -name '*kernel_trace.csv' -delete 2>/dev/null; find gpurun_out -name '*agent_info.csv' -delete 2>/dev/null
