#!/usr/bin/env bash
# Round 5, GPU call 38: the final tree again after T = 88 (M = 16, k <= 16) -- full GPU suite, smoke, the default bench line, rocprofv3 kernel
# command, the shard pair behind the 8-GPU estimate
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; OUT=gpurun_out/r05c38; mkdir -p $OUT gpurun_out/r05p; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_final.txt 2>&1; echo "suite rc=$?"; tail -3 $OUT/pytest_gpu_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
S=$(date +%s); timeout 900 python bench.py > $OUT/bench_10m_n1_final.json 2>$OUT/bench_10m_n1_final.err; echo "bench rc=$? wall $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05c38/bench_10m_n1_final.json') if l.startswith('{')][-1]); r=d['roofline']; c=d['cpu_baseline']
print('bench: %.0f q/s  %.4f ms/step  kernel %.4f  frac %.3f (at measured clock %s, %s MHz)  traffic %s  parity %s (%s queries)  sha %s' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('frac_at_measured_clock'), r.get('shader_clock_mhz'), r.get('traffic'), c.get('gpu_matches_cpu_bit_exact_all'), c.get('queries_checked'), d['result_sha256'][:16]))
for k,v in d.get('legs', {}).items():
    if isinstance(v, dict): print('  leg %-8s %s' % (k, {kk: v[kk] for kk in ('value','ms_per_step','recall_at_10') if kk in v}))
PY
bash scripts/r05_profiles.sh stats shards 2>&1 | tail -16
cp gpurun_out/r05p/* $OUT/ 2>/dev/null; rm -rf gpurun_out/r05p/bench_trace; du -sh gpurun_out
