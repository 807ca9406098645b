#!/usr/bin/env bash
# Round 6, call 10: how often the scanning waves pick up the workgroup's bounds (every 2nd / 4th / 8th step), alternating on one box;
# the re-rank leg's pool size (candidates per slice) against recall and q/s.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c10; mkdir -p $OUT
A="--legs none --cpu-queries 0 --recall-queries 0 --no-rerank --warmup 20"
for lib in base thw3 thw7 base thw3 thw7; do
  E=""; [ $lib != base ] && E="ANNLITE_HIP_LIB=$PWD/annlite_amd/libannlite_hip_$lib.so"
  for cfg in "10m --steps 100" "1250000 --rows 1250000 --steps 200 --streams 2" "1000000 --rows 1000000 --steps 200 --streams 2" "k50 --k 50 --steps 50 --streams 2"; do
    tag=${cfg%% *}; args=${cfg#* }
    env $E timeout 200 python bench.py $A $args 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('%-5s %-8s q/s %9.0f ms/step %.4f kernel_ms %.4f frac %.3f at-clock %.3f (%s MHz) sha %s' % ('$lib', '$tag', d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('frac_at_measured_clock') or 0, int(r.get('shader_clock_mhz') or 0), d['result_sha256'][:8]))"
  done
done 2>&1 | tee $OUT/thw_mask_ab.txt
for rk in 8 10 12 14 16; do
  timeout 300 python bench.py --legs rerank --rerank-k $rk --cpu-queries 0 --steps 40 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['rerank']
print('rerank_k %2d (pool 8 x %2d = %3d rows): %9.0f q/s at recall@10 %.4f' % ($rk, $rk, 8 * $rk, r['value'], r['recall_at_10']))"
done 2>&1 | tee $OUT/rerank_pool_sweep.txt
