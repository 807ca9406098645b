#!/usr/bin/env bash
# round 3, last build: the full GPU suite and the default bench line once more (kernel-choice policy changed after r03_final.sh)
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03final2; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
t1=$(date +%s); echo "pytest -m gpu: $((t1-t0)) s" | tee -a $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
t0=$(date +%s)
timeout 900 python bench.py > $OUT/bench_10m_n1.json 2> $OUT/bench_10m_n1.err
t1=$(date +%s); echo "default bench.py: $((t1-t0)) s"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03final2/bench_10m_n1.json')); r=d['roofline']
print('default: ms/step %.4f kernel %.4f frac %.3f q/s %.0f rerank %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], d.get('rerank') and (d['rerank'].get('value'), d['rerank'].get('recall_at_10'))))
for k in ('c2','c4','c5','uniform'):
    v=d.get(k) or {}
    print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('kernel_choice'), v.get('recall_at_10'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['gpu_matches_cpu_bit_exact'])
PY
