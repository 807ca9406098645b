#!/usr/bin/env bash
# Round 5, GPU call 26: seed rows scaled with k for the 64-key lists: tests + timings (k = 50 / 64 / 20 / 17), the k50 bench leg's own command
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; OUT=gpurun_out/r05c26; mkdir -p $OUT
timeout 600 python -m pytest tests/test_k64_byte_tables.py tests/test_k64_stress.py -x -q > $OUT/pytest_k64.txt 2>&1; echo "k64 rc=$?"; tail -3 $OUT/pytest_k64.txt
P="--rows 10000000 --data lowrank --fused --valid --iters 10"
for k in 50 64 20 17; do
  echo "k=$k: $(ANNLITE_SCAN_VARIANT=50 timeout 90 python scripts/prof_scan.py $P --k $k 2>&1 | grep -v '^/opt' | head -3 | tr '\n' ' ' | cut -c1-200)"
done
echo "k=50 at 1.25M rows: $(ANNLITE_SCAN_VARIANT=50 timeout 90 python scripts/prof_scan.py --rows 1250000 --data lowrank --fused --valid --iters 16 --k 50 2>&1 | grep -v '^/opt' | head -3 | tr '\n' ' ' | cut -c1-200)"
echo "k=50 at 1.25M rows, 32768 seed rows: $(ANNLITE_SEED_ROWS=32768 ANNLITE_SCAN_VARIANT=50 timeout 90 python scripts/prof_scan.py --rows 1250000 --data lowrank --fused --valid --iters 16 --k 50 2>&1 | grep -v '^/opt' | head -3 | tr '\n' ' ' | cut -c1-200)"
timeout 300 python bench.py --k 50 --legs none --steps 20 --warmup 5 --cpu-queries 16 --cpu-repeats 3 --recall-queries 32 --streams 2 --query-batches 2 > $OUT/bench_k50_leg.json 2>/dev/null; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05c26/bench_k50_leg.json') if l.startswith('{')][-1]); r=d['roofline']; c=d['cpu_baseline']
print('k50 leg: %.0f q/s  %.4f ms/step  kernel %.4f  frac %.3f  clock %s  parity %s (%s queries)' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['shader_clock_mhz'], c['gpu_matches_cpu_bit_exact_all'], c['queries_checked']))
PY
