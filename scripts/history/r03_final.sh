#!/usr/bin/env bash
# round 3, final build: full GPU suite, the default bench line (all legs), then scripts/r03_profiles.sh
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03final; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
t1=$(date +%s); echo "pytest -m gpu: $((t1-t0)) s" | tee -a $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
t0=$(date +%s)
timeout 900 python bench.py > $OUT/bench_10m_n1.json 2> $OUT/bench_10m_n1.err
t1=$(date +%s); echo "default bench.py: $((t1-t0)) s"
C4="--dim 768 --m 64 --batch 256 --metric cosine --legs none --cpu-queries 8 --cpu-repeats 1 --recall-queries 64"
timeout 600 python bench.py --rows 10000000 $C4 > $OUT/bench_config4_10m_n1.json 2>/dev/null
timeout 600 python bench.py --rows 2000000 $C4 > $OUT/bench_config4_2m_n1.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03final/bench_10m_n1.json')); r=d['roofline']
print('default: ms/step %.4f kernel %.4f frac %.3f q/s %.0f rerank %s legs %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], d.get('rerank') and (d['rerank'].get('value'), d['rerank'].get('recall_at_10')), {k:(v.get('value') if isinstance(v,dict) else v) for k,v in (d.get('legs') or {}).items()}))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['gpu_matches_cpu_bit_exact'])
for f in ('bench_config4_10m_n1','bench_config4_2m_n1'):
    d=json.load(open('gpurun_out/r03final/%s.json'%f)); r=d['roofline']
    print(f, 'ms/step %.4f kernel %.4f frac %.3f q/s %.0f rerank %s parity %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], d.get('rerank') and (d['rerank'].get('value'), d['rerank'].get('recall_at_10')), d['cpu_baseline'] and d['cpu_baseline']['gpu_matches_cpu_bit_exact']))
PY
bash scripts/r03_profiles.sh 2>&1 | tail -40
