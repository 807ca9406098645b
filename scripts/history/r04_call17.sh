#!/usr/bin/env bash
# round 4, GPU call 17: VERDICT r3 item 1c -- one GPU's share of a 2 query-groups x 4 row-shards grid (2.5M rows x 512 queries)
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r04v; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
A="--rows 2500000 --batch 512 --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 200 --warmup 20"
timeout 150 python bench.py $A --streams 2 > $OUT/grid2x4_s2.json 2>/dev/null
ANNLITE_FORCE_GATHER=1 timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 $A --streams 2 > $OUT/grid2x4_forced_gather.json 2>/dev/null
B="--rows 1250000 --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 200 --warmup 20"
ANNLITE_FORCE_GATHER=1 timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 1 $B --streams 2 > $OUT/rowshard8_forced_gather.json 2>/dev/null
timeout 150 python bench.py --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 100 --warmup 20 > $OUT/one_gpu_s1.json 2>/dev/null
python - <<'PY' | tee gpurun_out/r04v/grid_table.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r04v/*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
    print('%-34s batch %d rows %d  ms/step %.4f  kernel_ms %.4f frac %.3f' % (f.split('/')[-1], d['config']['batch'], d['config']['rows_per_gpu'], d['ms_per_step'], r['kernel_ms'], r['frac']))
PY
