#!/usr/bin/env bash
# round 3, GPU call 3: candidate lists with per-slice bounds (parity + re-rank leg), loop experiments
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c4; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q  > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
P="python scripts/prof_scan.py --data lowrank --fused --iters 24 --rows 1250000"
for lib in e3 exp3; do
  export ANNLITE_HIP_LIB=$ROOT/build_exp/lib_$lib.so
  $P > $OUT/plain_${lib}_novalid.txt 2>&1
  $P --valid > $OUT/plain_${lib}_valid.txt 2>&1
  ANNLITE_DEBUG_SKIP=4 $P --valid > $OUT/skip4_${lib}_valid.txt 2>&1
  ANNLITE_DEBUG_SKIP=4 $P > $OUT/skip4_${lib}_novalid.txt 2>&1
  ANNLITE_DEBUG_COUNTERS=2 $P --valid > $OUT/timeline_${lib}_valid.txt 2>&1
  ANNLITE_DEBUG_COUNTERS=2 ANNLITE_DEBUG_SKIP=4 $P --valid > $OUT/timeline_skip4_${lib}_valid.txt 2>&1
done
unset ANNLITE_HIP_LIB
grep -H "scan kernel\|timeline" $OUT/*.txt | cut -c1-330
Q="--legs rerank --cpu-queries 0 --steps 20 --warmup 5"
python bench.py $Q --rerank-k 16 > $OUT/bench_rr16.json 2> $OUT/bench_rr16.err
ANNLITE_SCAN_SLICES=16 python bench.py $Q --rerank-k 16 > $OUT/bench_rr16_s16.json 2> $OUT/bench_rr16_s16.err

python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c4/bench_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], 'value %.0f ms %.4f' % (d['value'], d['ms_per_step']), 'rerank', d['rerank'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
