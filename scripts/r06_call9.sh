#!/usr/bin/env bash
# Round 6, call 9: a long randomised parity run on the final tree (new shapes: uint16 codes, Ks < 256, M = 24 / 128, scan state, the opt-in
# MFMA seed), then compile-time knobs of the step loop under the permute addressing (look-ups in flight, bound pick-up interval).
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c9; mkdir -p $OUT
timeout 420 python tests/fuzz_parity.py --seconds 360 --seed 61 2>&1 | grep -v "^/opt" | tail -12 | tee $OUT/fuzz_parity_seed61.txt
timeout 300 python tests/fuzz_parity.py --seconds 240 --seed 62 2>&1 | grep -v "^/opt" | tail -12 | tee $OUT/fuzz_parity_seed62.txt
A="--legs none --cpu-queries 0 --recall-queries 0 --no-rerank --warmup 20"
for lib in base d6 d10 d12 thw3 base; do
  E=""; [ $lib != base ] && E="ANNLITE_HIP_LIB=$PWD/annlite_amd/libannlite_hip_$lib.so"
  for cfg in "10m --steps 100" "1250000 --rows 1250000 --steps 200 --streams 2"; do
    tag=${cfg%% *}; args=${cfg#* }
    env $E timeout 200 python bench.py $A $args 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('%-5s %-8s q/s %9.0f ms/step %.4f kernel_ms %.4f frac %.3f at-clock %.3f (%s MHz) sha %s' % ('$lib', '$tag', d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('frac_at_measured_clock') or 0, int(r.get('shader_clock_mhz') or 0), d['result_sha256'][:8]))"
  done
done 2>&1 | tee $OUT/step_loop_knobs.txt
