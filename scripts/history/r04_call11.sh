#!/usr/bin/env bash
# round 4, GPU call 11: full suite; the shard with the exchange forced: plain vs opt-in seed exchange with 8 emulated peers
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r04c11; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.txt
tail -8 $OUT/pytest.txt
A="--rows 1250000 --legs none --cpu-queries 4 --cpu-repeats 1 --recall-queries 0 --no-rerank --steps 200 --warmup 20 --streams 2"
T="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
export ANNLITE_FORCE_GATHER=1
p=29550
for v in "" "--seed-exchange" "--seed-exchange --emulate-seed-peers 8" "" "--seed-exchange --emulate-seed-peers 8"; do
  p=$((p+1)); n=$(echo "p${p}_$v" | tr -d ' -')
  timeout 300 $T --master-port $p bench.py --gpus 1 $A $v > $OUT/shard_$n.json 2>$OUT/err_$n.txt
done
unset ANNLITE_FORCE_GATHER
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04c11/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']; c=d['config']
        print('%-52s ms/step %.4f host %.4f kernel_ms %.4f seedx %s rows %s emul %s parity %s' % (f.split('/')[-1], d['ms_per_step'], d['host_enqueue_ms_per_step'], r['kernel_ms'], c.get('seed_exchange'), c.get('seed_rows'), c.get('seed_peers_emulated'), (d.get('cpu_baseline') or {}).get('gpu_matches_cpu_bit_exact_all')))
    except Exception as e: print(f, 'ERR', e)
PY
