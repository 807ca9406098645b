#!/usr/bin/env bash
# Round 6, call 2: the MFMA-nominated seed -- its own tests first, then the whole GPU suite, then A/B against the exact seed scan
# (ANNLITE_NO_MFMA_SEED=1) at the shard / config-2 / headline sizes, digests compared.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c2; mkdir -p $OUT
timeout 600 python -m pytest tests/test_seed_mfma.py -x -q > $OUT/pytest_seed_mfma.txt 2>&1; echo "seed_mfma rc=$?"; tail -15 $OUT/pytest_seed_mfma.txt
timeout 900 python -m pytest tests -q -m gpu --maxfail=8 --deselect tests/test_seed_mfma.py > $OUT/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -8 $OUT/pytest_gpu.txt
A="--legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 200 --warmup 20"
for mode in mfma exact; do
  E=""; [ $mode = exact ] && E="ANNLITE_NO_MFMA_SEED=1"
  env $E timeout 120 python bench.py --rows 1250000 $A --streams 2 > $OUT/bench_shard_1250000_s2_$mode.json 2>/dev/null
  env $E ANNLITE_FORCE_GATHER=1 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --rows 1250000 $A --streams 2 > $OUT/bench_shard_1250000_forced_gather_$mode.json 2>/dev/null
  env $E timeout 120 python bench.py --rows 1000000 $A --streams 2 > $OUT/bench_c2_s2_$mode.json 2>/dev/null
  env $E timeout 120 python bench.py --rows 1000000 $A --streams 1 > $OUT/bench_c2_s1_$mode.json 2>/dev/null
  env $E timeout 200 python bench.py --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 100 --warmup 10 > $OUT/bench_10m_$mode.json 2>/dev/null
  env $E timeout 200 python bench.py --k 50 --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 50 --warmup 10 --streams 2 > $OUT/bench_10m_k50_$mode.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06c2/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1]); r = d['roofline']
        print('%-50s q/s %9.0f  ms/step %.4f  kernel_ms %.4f  frac %.3f  streams %s sha %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], d['config'].get('streams'), d['result_sha256'][:8]))
    except Exception as e:
        print(f, 'ERR', e)
PY
# where the preparation launch's time goes now (stamps of its first / last workgroup) and the nomination launch's own duration
cat > /tmp/prep_tl.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from annlite_amd import ops, _capi
from annlite_amd._capi import LUT_L2
rs = np.random.RandomState(1)
N, B = 1250000, 1024
A = rs.randn(16, 128).astype(np.float32)
cb = ops.to_dev((rs.randn(256, 16).astype(np.float32) @ A).reshape(256, 16, 8).transpose(1, 0, 2).copy())
x = torch.from_numpy((rs.randn(N, 16).astype(np.float32) @ A + 0.05 * rs.randn(N, 128).astype(np.float32))).cuda()
codes = ops.codes_skew(ops.pq_encode(x, cb))
q = ops.to_dev((rs.randn(B, 16).astype(np.float32) @ A).astype(np.float32))
st = _capi.ScanState()
for _ in range(5): ops.pq_search_topk(LUT_L2, q, cb, codes, 10, 16, 256, codes_layout=1, state=st)
torch.cuda.synchronize()
print(os.environ.get('ANNLITE_NO_MFMA_SEED', 'mfma'), _capi.debug_prep_timeline())
cand = ops.debug_seed_candidates(q, cb, codes, 40960, codes_layout=1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for S in (8192, 32768, 131072):
    for _ in range(3): ops.debug_seed_candidates(q, cb, codes, S, codes_layout=1)
    e0.record()
    for _ in range(20): ops.debug_seed_candidates(q, cb, codes, S, codes_layout=1)
    e1.record(); torch.cuda.synchronize()
    print('nomination launch, S = %6d: %.2f us per launch (back to back, 1024 queries)' % (S, e0.elapsed_time(e1) / 20 * 1e3))
PY
ANNLITE_DEBUG_COUNTERS=2 timeout 120 python /tmp/prep_tl.py 2>&1 | grep -v "^/opt" | tee $OUT/prep_timeline_mfma.txt
ANNLITE_DEBUG_COUNTERS=2 ANNLITE_NO_MFMA_SEED=1 timeout 120 python /tmp/prep_tl.py 2>&1 | grep -v "^/opt" | head -1 | tee $OUT/prep_timeline_exact.txt
