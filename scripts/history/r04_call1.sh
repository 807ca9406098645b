#!/usr/bin/env bash
# round 4, GPU call 1: the whole -m gpu suite on the new build (non-finite inputs, lazy matches, prebuilt byte tables, parallel seed
# selection), then the 1.25M-row shard with and without the prebuilt tables, and the preparation launch's phase stamps
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r04c1; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.txt
tail -45 $OUT/pytest.txt
A="--rows 1250000 --legs none --cpu-queries 8 --cpu-repeats 1 --recall-queries 0 --no-rerank --steps 200 --warmup 20"
for s in 2 1; do
  timeout 200 python bench.py $A --streams $s > $OUT/shard_prebuilt_s$s.json 2>$OUT/err_a$s.txt
  ANNLITE_NO_PREBUILT_TABLES=1 timeout 200 python bench.py $A --streams $s > $OUT/shard_noprebuilt_s$s.json 2>$OUT/err_b$s.txt
done
ANNLITE_DEBUG_COUNTERS=2 timeout 200 python scripts/prof_scan.py --rows 1250000 --data lowrank --fused --valid --iters 30 > $OUT/prep_timeline.txt 2>&1
ANNLITE_NO_PREBUILT_TABLES=1 ANNLITE_DEBUG_COUNTERS=2 timeout 200 python scripts/prof_scan.py --rows 1250000 --data lowrank --fused --valid --iters 30 > $OUT/prep_timeline_noprebuilt.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04c1/shard_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f kernel_ms %.4f frac %.3f parity %s/%s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['cpu_baseline']['gpu_matches_cpu_bit_exact'], d['cpu_baseline'].get('gpu_matches_cpu_bit_exact_all')))
    except Exception as e: print(f, 'ERR', e)
PY
tail -12 $OUT/prep_timeline.txt; tail -6 $OUT/prep_timeline_noprebuilt.txt
