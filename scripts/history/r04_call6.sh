#!/usr/bin/env bash
# round 4, GPU call 6: the split search (PREPARE / seed union / SCAN), bench with the exchange forced: plain, with the seed exchange
# (one rank), and with 8 emulated seed peers -- the 1.25M-row shard of an 8-rank job
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r04c6; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_round4_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "split_search or seed_exchange" > $OUT/pytest_new.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_new.txt
tail -25 $OUT/pytest_new.txt
A="--rows 1250000 --legs none --cpu-queries 4 --cpu-repeats 1 --recall-queries 0 --no-rerank --steps 200 --warmup 20 --streams 2"
T="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
export ANNLITE_FORCE_GATHER=1
timeout 300 $T --master-port 29541 bench.py --gpus 1 $A --no-seed-exchange > $OUT/shard_gather_noseedx.json 2>$OUT/err1.txt
timeout 300 $T --master-port 29542 bench.py --gpus 1 $A > $OUT/shard_gather_seedx_1rank.json 2>$OUT/err2.txt
timeout 300 $T --master-port 29543 bench.py --gpus 1 $A --emulate-seed-peers 8 > $OUT/shard_gather_seedx_emul8.json 2>$OUT/err3.txt
timeout 300 $T --master-port 29544 bench.py --gpus 1 $A --no-seed-exchange > $OUT/shard_gather_noseedx_again.json 2>$OUT/err4.txt
timeout 300 $T --master-port 29545 bench.py --gpus 1 $A --emulate-seed-peers 8 > $OUT/shard_gather_seedx_emul8_again.json 2>$OUT/err5.txt
unset ANNLITE_FORCE_GATHER
timeout 300 python bench.py --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 100 --warmup 20 > $OUT/bench_10m_s1.json 2>$OUT/err_10m.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04c6/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']; c=d['config']
        print(f.split('/')[-1], 'ms/step %.4f q/s %.0f kernel_ms %.4f exch %s seedx %s rows %s emul %s parity %s' % (d['ms_per_step'], d['value'], r['kernel_ms'], d.get('exchange_ms'), c.get('seed_exchange'), c.get('seed_rows'), c.get('seed_peers_emulated'), (d.get('cpu_baseline') or {}).get('gpu_matches_cpu_bit_exact_all')))
    except Exception as e: print(f, 'ERR', e)
PY
tail -3 $OUT/err3.txt
