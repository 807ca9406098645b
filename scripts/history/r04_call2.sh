#!/usr/bin/env bash
# round 4, GPU call 2: the -m gpu suite again (large-k fix), then the shard on 1..4 streams and with the exchange forced
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r04c2; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.txt
tail -60 $OUT/pytest.txt
A="--rows 1250000 --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 200 --warmup 20"
for s in 2 3 4; do timeout 200 python bench.py $A --streams $s > $OUT/shard_s$s.json 2>$OUT/err_s$s.txt; done
for s in 2 3; do ANNLITE_FORCE_GATHER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$s bench.py --gpus 1 $A --streams $s > $OUT/shard_gather_s$s.json 2>$OUT/err_g$s.txt; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04c2/shard_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f kernel_ms %.4f frac %.3f exch %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d.get('exchange_ms')))
    except Exception as e: print(f, 'ERR', e)
PY
