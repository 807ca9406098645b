#!/usr/bin/env bash
# First GPU call of round 6: the GPU suite on the refactored tree (switches parsed at load, the two-rank bench test), the default
# bench line, and the small-table pair the round works on (1.25M-row shard with the exchange forced; 1M rows = config 2), each
# step under its own timeout.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06first; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -5 $OUT/pytest_gpu.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_10m_n1.json 2>$OUT/bench_10m_n1.err; echo "bench rc=$?"; tail -c 1500 $OUT/bench_10m_n1.json; echo
A="--legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 200 --warmup 20"
timeout 120 python bench.py --rows 1250000 $A --streams 2 > $OUT/bench_shard_1250000_s2.json 2>/dev/null
ANNLITE_FORCE_GATHER=1 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --rows 1250000 $A --streams 2 > $OUT/bench_shard_1250000_forced_gather.json 2>/dev/null
timeout 120 python bench.py --rows 1000000 $A --streams 2 > $OUT/bench_c2_s2.json 2>/dev/null
timeout 120 python bench.py --rows 1000000 $A --streams 1 > $OUT/bench_c2_s1.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06first/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1]); r = d['roofline']
        print('%-46s q/s %9.0f  ms/step %.4f  kernel_ms %.4f  frac %.3f  streams %s host_enq %.4f sha %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], d['config'].get('streams'), d['host_enqueue_ms_per_step'], d['result_sha256'][:8]))
    except Exception as e:
        print(f, 'ERR', e)
PY
