#!/usr/bin/env bash
# Round 5, GPU call 34: interleaved row slices (ANNLITE_Q8_ILV = log2 of the run length in blocks; 0 = contiguous slices):
# exactness subset, then A/B on the bench's tables and on a table in cluster order
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; OUT=gpurun_out/r05c34; mkdir -p $OUT
timeout 900 python -m pytest tests/test_seed_rows_spread.py tests/test_gpu_parity.py tests/test_round4_gpu.py tests/test_k64_byte_tables.py tests/test_k64_stress.py tests/test_m32_byte_tables.py -m gpu -x -q > $OUT/pytest_subset.txt 2>&1
echo "subset rc=$?"; tail -4 $OUT/pytest_subset.txt
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']
    print('ms/step %.4f  q/s %.0f  kernel_ms %.4f  sha %s' % (d['ms_per_step'], d['value'], r['kernel_ms'], d['result_sha256'][:10]))
except Exception as e: print('ERR', e)
PY
}
C="--legs none --cpu-queries 0 --recall-queries 0 --no-rerank --streams 2 --steps 200 --warmup 20"
for rows in 1250000 10000000; do
  for V in 4 0 2 6 4 0; do
    ANNLITE_Q8_ILV=$V timeout 200 python bench.py --rows $rows $C > $OUT/t_${rows}_ilv$V.json 2>/dev/null; echo "rows $rows ilv $V: $(line $OUT/t_${rows}_ilv$V.json)"
  done
done
for rows in 1250000 10000000; do
  for V in 4 0 2 6; do
    echo "sorted rows $rows ilv $V: $(ANNLITE_Q8_ILV=$V timeout 200 python scripts/prof_scan.py --rows $rows --fused --data lowrank --order sorted --iters 12 2>&1 | grep -E 'whole call|scan kernel ms' | tr '\n' ' ')"
  done
done
for V in 4 0; do
  echo "m32 10M ilv $V: $(ANNLITE_Q8_ILV=$V timeout 200 python scripts/prof_scan.py --rows 10000000 --m 32 --dsub 4 --fused --data lowrank --iters 10 2>&1 | grep -E 'whole call|scan kernel ms' | tr '\n' ' ')"
  echo "m64 10M b256 ilv $V: $(ANNLITE_Q8_ILV=$V timeout 200 python scripts/prof_scan.py --rows 10000000 --m 64 --dsub 2 --batch 256 --fused --data lowrank --iters 10 2>&1 | grep -E 'whole call|scan kernel ms' | tr '\n' ' ')"
  echo "k50 10M ilv $V: $(ANNLITE_Q8_ILV=$V timeout 200 python scripts/prof_scan.py --rows 10000000 --k 50 --fused --data lowrank --iters 10 2>&1 | grep -E 'whole call|scan kernel ms' | tr '\n' ' ')"
done
