#!/usr/bin/env bash
# Round 6, call 33: seeds of the insertion walks / searches on the hard shape (500k x 768, m = 64).
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c33; mkdir -p $OUT
for sd in 128 1024 32; do
  timeout 600 python scripts/bench_hnsw.py --rows 500000 --dim 768 --m 64 --batch 256 --steps 10 --build gpu --gpu-seeds $sd 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('seeds $sd: build_s %.2f  adc recall %.4f  re-rank recall %.4f  %.0f q/s' % (d['build_s'], d['hnsw_gpu_walk_adc']['recall_at_10'], d['recall_at_10'], d['value']))"
done | tee $OUT/seeds_m64.txt
