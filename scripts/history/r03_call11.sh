#!/usr/bin/env bash
# round 3, GPU call 11: full GPU suite (C5 at 5M rows, hardened parity tests) + the default bench run with every leg
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c11; mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s); timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - T0 )) s"; tail -4 $OUT/pytest.log
T0=$(date +%s); python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default: $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r03c11/bench_default.json'))
    r=d['roofline']
    print('main: ms/step %.4f kernel %.4f frac %.3f q/s %.0f recall %.3f parity %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], d['recall_at_10'], d['cpu_baseline']['gpu_matches_cpu_bit_exact']))
    print('cpu:', {k:(v if not isinstance(v,dict) else v.get('value')) for k,v in d['cpu_baseline'].items()})
    print('rerank:', d['rerank']); print('ivf:', d['ivf'])
    for leg in ('c2','c4','c5','uniform'):
        v=d.get(leg)
        if v is None: print(leg, 'MISSING'); continue
        if 'error' in v: print(leg, v); continue
        print(leg, v['config'], '| value %.0f' % v['value'], '| recall', v.get('recall_at_10'), '| roof', v['roofline'].get('frac'), v['roofline'].get('kernel'), v['roofline'].get('kernel_ms'), '| cpu', v['cpu_baseline'] and v['cpu_baseline'].get('value'))
except Exception as e:
    print('ERR', e); print(open('gpurun_out/r03c11/bench_default.err').read()[-1500:])
PY
