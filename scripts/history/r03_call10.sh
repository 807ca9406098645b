#!/usr/bin/env bash
# round 3, GPU call 7: library-side kernel choice (guard + gated fallback + state), rank-counting tile merge
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c10; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
P="python scripts/prof_scan.py --data lowrank --fused --iters 24 --valid"
Q="--legs none --cpu-queries 8 --cpu-repeats 1 --steps 60 --warmup 10"
ANNLITE_DEBUG_COUNTERS=2 $P --rows 1250000 > $OUT/items_1p25m.txt 2>&1
ANNLITE_DEBUG_COUNTERS=2 $P --rows 10000000 --iters 10 > $OUT/items_10m.txt 2>&1
grep -H "scan kernel\|timeline\|items:\|last to end\|merge phase\|kernel choice" $OUT/items*.txt | cut -c1-330
for st in 1 2; do python bench.py --rows 1250000 $Q --streams $st > $OUT/bench_1p25m_s$st.json 2>$OUT/bench_1p25m_s$st.err; done
python bench.py $Q --steps 30 > $OUT/bench_10m.json 2>$OUT/bench_10m.err
python bench.py $Q --steps 10 --warmup 3 --data uniform > $OUT/bench_10m_uniform.json 2>$OUT/bench_10m_uniform.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c10/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f  parity %s  %s %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], d['cpu_baseline'] and d['cpu_baseline']['gpu_matches_cpu_bit_exact'], r['kernel'], r.get('kernel_choice')))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
PY
python - <<'PY'
# per-batch cost of the stateless (always guarded) call against the call with a state, structured data
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from annlite_amd import ops, _capi, Metric, PQCodec
from annlite_amd._capi import LUT_L2
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(0)
N, M, Ks, B, k, D = 1_250_000, 16, 256, 1024, 10, 128
A = torch.randn((16, D), generator=g, device=dev)
gen = lambda n: (torch.randn((n, 16), generator=g, device=dev) @ A + 0.05 * torch.randn((n, D), generator=g, device=dev)).contiguous()
codec = PQCodec(dim=D, n_subvectors=M, n_clusters=Ks, metric=Metric.EUCLIDEAN, n_init=1); codec.seed = 7
codec.fit(gen(20480), iter=20)
cb = codec.codebooks_dev
codes = ops.codes_skew(ops.pq_encode(gen(N), cb)); q = gen(B)
ws = ops.ScanWorkspace(); st = _capi.ScanState()
def t(state, n=100):
    for _ in range(10): ops.pq_search_topk(LUT_L2, q, cb, codes, k, M, Ks, codes_layout=1, workspace=ws, state=state)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ops.pq_search_topk(LUT_L2, q, cb, codes, k, M, Ks, codes_layout=1, workspace=ws, state=state)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('1.25M rows, ms per batch: with state %.4f (%s), stateless (guarded + gated pass) %.4f' % (t(st), st.info(), t(None)))
PY
cat gpurun_out/kernel_choice_2m_uniform_codes.txt
