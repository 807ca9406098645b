#!/usr/bin/env bash
# Round 6, call 6: permute addressing with the LDS size fixed + the LDS-table generic kernel: GPU suite (per-test timeout), then the
# same-box A/B against the previous tree's library (k = 50 digest included) and the M = 128 / M = 64 numbers.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c6; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu --maxfail=10 --timeout 400 > $OUT/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -12 $OUT/pytest_gpu.txt
A="--legs none --cpu-queries 0 --recall-queries 0 --no-rerank --warmup 20"
for lib in new prev new prev; do
  E=""; [ $lib = prev ] && E="ANNLITE_HIP_LIB=$PWD/annlite_amd/libannlite_hip_prev.so"
  for cfg in "10m --steps 100" "1250000 --rows 1250000 --steps 200 --streams 2" "1000000 --rows 1000000 --steps 200 --streams 2" "k50 --k 50 --steps 50 --streams 2"; do
    tag=${cfg%% *}; args=${cfg#* }
    env $E timeout 200 python bench.py $A $args 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('%-5s %-8s q/s %9.0f ms/step %.4f kernel_ms %.4f frac %.3f at-clock %.3f (%s MHz) sha %s' % ('$lib', '$tag', d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('frac_at_measured_clock') or 0, int(r.get('shader_clock_mhz') or 0), d['result_sha256'][:8]))"
  done
done 2>&1 | tee $OUT/perm_addressing_ab.txt
for shape in "--m 128 --dsub 1" "--m 64 --dsub 2"; do
  tag=$(echo $shape | tr -d ' -')
  timeout 200 python scripts/prof_scan.py --rows 1000000 --batch 256 --data lowrank --fused --valid --iters 6 --layout 0 $shape 2>&1 | grep -v "^/opt" | tail -3 > $OUT/scan_1m_${tag}_plain.txt
  echo "== $shape"; cat $OUT/scan_1m_${tag}_plain.txt
done
