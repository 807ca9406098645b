#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c33; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
A="--rows 1250000 --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 200 --warmup 20"
ANNLITE_FORCE_GATHER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 $A > $OUT/bench_1p25m_forced_gather.json 2> $OUT/err.txt || tail -5 $OUT/err.txt
timeout 300 python bench.py $A --streams 2 > $OUT/bench_1p25m_s2.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c33/bench_*.json')):
    try:
        txt=[l for l in open(f) if l.startswith('{')][-1]
        d=json.loads(txt); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  q/s %.0f streams %s backend %s' % (d['ms_per_step'], r['kernel_ms'], d['value'], d['config'].get('streams'), d['config'].get('backend')))
    except Exception as e: print(f, 'ERR', e)
PY
