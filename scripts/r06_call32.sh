#!/usr/bin/env bash
# Round 6, call 32: batch size / growth bound of the GPU graph build against graph quality on the hard shape (500k x 768, m = 64).
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c32; mkdir -p $OUT
for cfg in "16384 4" "16384 16" "4096 16" "2048 32"; do
  set -- $cfg
  timeout 600 python scripts/bench_hnsw.py --rows 500000 --dim 768 --m 64 --batch 256 --steps 10 --build gpu --gpu-build-batch $1 --gpu-build-grow $2 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('batch $1 grow 1/$2: build_s %.2f  adc recall %.4f  re-rank recall %.4f  %.0f q/s' % (d['build_s'], d['hnsw_gpu_walk_adc']['recall_at_10'], d['recall_at_10'], d['value']))"
done | tee $OUT/build_batch_quality_m64.txt
for cfg in "16384 4" "4096 16"; do
  set -- $cfg
  timeout 600 python scripts/bench_hnsw.py --rows 2000000 --steps 10 --build gpu --gpu-build-batch $1 --gpu-build-grow $2 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('2M x 128 m16: batch $1 grow 1/$2: build_s %.2f  adc recall %.4f  re-rank recall %.4f  %.0f q/s' % (d['build_s'], d['hnsw_gpu_walk_adc']['recall_at_10'], d['recall_at_10'], d['value']))"
done | tee -a $OUT/build_batch_quality_m64.txt
