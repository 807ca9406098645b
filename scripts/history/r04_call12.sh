#!/usr/bin/env bash
# round 4, GPU call 12: the default bench line (all legs, facade included); shard knobs (slices, seed rows) on one box
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r04c12; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python bench.py > $OUT/bench_10m_n1.json 2>$OUT/bench_err.txt ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04c12/bench_10m_n1.json') if l.startswith('{')][-1])
print('value %.0f ms/step %.4f kernel %.4f frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))
print('rerank', d['rerank']); print('facade', json.dumps(d['facade'])[:900]); print('ivf', d['ivf'])
for k in ('c2','c4','c5','uniform'):
    r=d.get(k) or {}
    print(k, r.get('value'), r.get('ms_per_step'), (r.get('roofline') or {}).get('frac'), r.get('recall_at_10'), r.get('error'))
cb=d['cpu_baseline']; print('cpu', cb['value'], cb['all_cores'], cb['gpu_matches_cpu_bit_exact'], cb['gpu_matches_cpu_bit_exact_all'], cb['queries_checked'], cb['queries_differing'])
PY
timeout 300 python scripts/ab_scan.py --rows 1250000 --envs "|ANNLITE_SCAN_SLICES=16|ANNLITE_SEED_ROWS=16384|ANNLITE_SEED_ROWS=65536||ANNLITE_SCAN_SLICES=16,ANNLITE_SEED_ROWS=16384" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_shard_knobs.txt
