#!/usr/bin/env bash
# round 4, GPU call 14: the shard table behind the multi-GPU estimate (two streams; two streams + the exchange forced on one rank)
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r04s; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
p=29520
for rows in 10000000 5000000 2500000 1250000; do
  A="--rows $rows --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 200 --warmup 20"
  timeout 150 python bench.py $A --streams 1 > $OUT/bench_shard_${rows}_s1_200steps.json 2>/dev/null
  timeout 150 python bench.py $A --streams 2 > $OUT/bench_shard_${rows}_s2_200steps.json 2>/dev/null
  p=$((p+1)); ANNLITE_FORCE_GATHER=1 timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 1 $A --streams 2 > $OUT/bench_shard_${rows}_forced_gather_200steps.json 2>/dev/null
done
python - <<'PY' | tee gpurun_out/r04s/shard_table.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r04s/bench_shard_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
        print('%-52s ms/step %.4f  kernel_ms %.4f frac %.3f q/s %.0f streams %s exchange_ms %s' % (f.split('/')[-1], d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], d['config'].get('streams'), d.get('exchange_ms')))
    except Exception as e: print(f, 'ERR', e)
PY
