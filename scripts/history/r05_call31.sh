#!/usr/bin/env bash
# Round 5, GPU call 31: seed rows spread over the table + first epoch end at step 255 -- exactness, A/B against the
# first-rows seed and the early epoch end, and a table filled in cluster order (the case the spread seed is for)
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; OUT=gpurun_out/r05c31; mkdir -p $OUT
timeout 600 python -m pytest tests/test_round4_gpu.py tests/test_k64_byte_tables.py tests/test_k64_stress.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_subset.txt 2>&1
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
for rows in 1250000 1000000 10000000; do
  for V in "new" "contig" "old_epoch" "contig_old_epoch" "new"; do
    case $V in
      new) E="";; contig) E="ANNLITE_SEED_CONTIGUOUS=1";; old_epoch) E="ANNLITE_Q8_TUNE=15,16,384,3";;
      contig_old_epoch) E="ANNLITE_SEED_CONTIGUOUS=1 ANNLITE_Q8_TUNE=15,16,384,3";;
    esac
    env $E timeout 200 python bench.py --rows $rows $C > $OUT/t_${rows}_$V.json 2>/dev/null; echo "rows $rows $V: $(line $OUT/t_${rows}_$V.json)"
  done
done
for V in "new" "old_epoch" "contig_old_epoch"; do
  case $V in new) E="";; old_epoch) E="ANNLITE_Q8_TUNE=15,16,384,7";; contig_old_epoch) E="ANNLITE_SEED_CONTIGUOUS=1 ANNLITE_Q8_TUNE=15,16,384,7";; esac
  env $E timeout 200 python bench.py --k 50 $C > $OUT/t_k50_$V.json 2>/dev/null; echo "k50 10M $V: $(line $OUT/t_k50_$V.json)"
done
# a table filled in cluster order: 1.25M and 10M rows
for rows in 1250000 10000000; do
  for V in "new" "contig" "contig_old_epoch" "old_epoch"; do
    case $V in
      new) E="";; contig) E="ANNLITE_SEED_CONTIGUOUS=1";; old_epoch) E="ANNLITE_Q8_TUNE=15,16,384,3";;
      contig_old_epoch) E="ANNLITE_SEED_CONTIGUOUS=1 ANNLITE_Q8_TUNE=15,16,384,3";;
    esac
    echo "sorted rows $rows $V: $(env $E timeout 200 python scripts/prof_scan.py --rows $rows --fused --data lowrank --order sorted --iters 12 2>&1 | grep -E 'whole call|scan kernel ms' | tr '\n' ' ')"
  done
done
