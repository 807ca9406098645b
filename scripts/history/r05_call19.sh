#!/usr/bin/env bash
# Round 5, GPU call 19: M = 32 byte-table plan with 8 slices + slice-per-XCD map at >= 8M rows: tests, the m32 leg's own command (parity of every query), timing.
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r05c19; rm -rf gpurun_out/*; mkdir -p $OUT
timeout 400 python -m pytest tests/test_m32_byte_tables.py tests/test_m32_addressing.py -x -q > $OUT/pytest_m32.txt 2>&1; echo "m32 tests rc=$?"; tail -3 $OUT/pytest_m32.txt
timeout 300 python bench.py --m 32 --legs none --steps 20 --warmup 5 --cpu-queries 16 --cpu-repeats 3 --recall-queries 32 --streams 2 --query-batches 2 > $OUT/bench_m32_leg.json 2>/dev/null; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05c19/bench_m32_leg.json') if l.startswith('{')][-1]); r=d['roofline']; c=d['cpu_baseline']
print('m32 leg: %.0f q/s  %.4f ms/step  kernel %.4f  frac %.3f  clock %s  at clock %s  parity %s (%s queries)  rows/gpu %s' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['shader_clock_mhz'], r['frac_at_measured_clock'], c['gpu_matches_cpu_bit_exact_all'], c['queries_checked'], d['config']['rows_per_gpu']))
PY
P="--rows 10000000 --m 32 --dsub 4 --data lowrank --fused --valid --iters 10"
timeout 90 python scripts/prof_scan.py $P 2>&1 | grep -v "^/opt" | head -3 | tr '\n' ' ' | cut -c1-220; echo
ANNLITE_SCAN_SLICES=4 timeout 90 python scripts/prof_scan.py $P 2>&1 | grep -v "^/opt" | head -3 | tr '\n' ' ' | cut -c1-220; echo
