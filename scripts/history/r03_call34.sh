#!/usr/bin/env bash
# the shard table of DESIGN section 8 at 200 timed steps (steady state), one and two streams, and with the exchange forced
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c34; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for rows in 10000000 5000000 2500000 1250000; do
  A="--rows $rows --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 200 --warmup 20"
  for s in 1 2; do timeout 300 python bench.py $A --streams $s > $OUT/bench_${rows}_s$s.json 2>/dev/null; done
  ANNLITE_FORCE_GATHER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 $A > $OUT/bench_${rows}_forced_gather.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c34/bench_*.json')):
    try:
        txt=[l for l in open(f) if l.startswith('{')][-1]
        d=json.loads(txt); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f frac %.3f q/s %.0f streams %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], d['config'].get('streams')))
    except Exception as e: print(f, 'ERR', e)
PY
