#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c15; mkdir -p $OUT
export TMPDIR=/tmp
python scripts/stress_fixture.py 10 | grep -v "^ 4\|first"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "m64 or config4 or fixture or random_shapes or ties or variants" 2>&1 | tail -2
C4="--dim 768 --m 64 --batch 256 --metric cosine --legs none --cpu-queries 8 --cpu-repeats 1 --recall-queries 32 --steps 20 --warmup 5"
timeout 300 python bench.py --rows 10000000 $C4 > $OUT/bench_c4_10m.json 2> $OUT/bench_c4_10m.err
for t in 256 512; do ANNLITE_Q8_TARGET=$t timeout 300 python bench.py --rows 10000000 $C4 --cpu-queries 0 > $OUT/bench_c4_10m_T$t.json 2>/dev/null; done
timeout 300 python bench.py --rows 1250000 $C4 --cpu-queries 0 > $OUT/bench_c4_1p25m.json 2>/dev/null
python scripts/bench_encode.py > $OUT/encode.jsonl 2>/dev/null; cat $OUT/encode.jsonl | cut -c1-400
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c15/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f  %s recall %.3f parity %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], r['kernel'], d['recall_at_10'], d['cpu_baseline'] and d['cpu_baseline']['gpu_matches_cpu_bit_exact']))
    except Exception as e: print(f, 'ERR', e)
PY
