#!/usr/bin/env bash
# round 4, GPU call 8: the shard with the exchange forced -- no seed exchange / seed exchange (1 rank) / 8 emulated peers (scans and
# preparation on the sharded index's own streams) -- and a kernel timeline of the emulated 8-rank pipeline
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r04c8; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_round4_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "split_search or seed_exchange" > $OUT/pytest_new.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_new.txt
A="--rows 1250000 --legs none --cpu-queries 4 --cpu-repeats 1 --recall-queries 0 --no-rerank --steps 200 --warmup 20"
T="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
export ANNLITE_FORCE_GATHER=1
p=29550
for rep in 1 2; do
  for v in "--no-seed-exchange" "" "--emulate-seed-peers 8"; do
    p=$((p+1)); n=$(echo "r${rep}_$v" | tr -d ' -')
    timeout 300 $T --master-port $p bench.py --gpus 1 $A --streams 2 $v > $OUT/shard_$n.json 2>$OUT/err_$n.txt
  done
done
bash scripts/gpu_timeline.sh seedx2 40 -- $T --master-port 29570 bench.py --gpus 1 --rows 1250000 --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 12 --warmup 4 --prewarm-steps 8 --streams 1 --emulate-seed-peers 8 > $OUT/timeline_seedx.txt 2>&1
unset ANNLITE_FORCE_GATHER
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob('gpurun_out/r04c8/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']; c=d['config']
        print('%-44s ms/step %.4f kernel_ms %.4f exch %.4f seedx %s rows %s emul %s parity %s' % (f.split('/')[-1], d['ms_per_step'], r['kernel_ms'], d.get('exchange_ms') or 0, c.get('seed_exchange'), c.get('seed_rows'), c.get('seed_peers_emulated'), (d.get('cpu_baseline') or {}).get('gpu_matches_cpu_bit_exact_all')))
    except Exception as e: print(f, 'ERR', e)
f=glob.glob('gpurun_out/tl_seedx2/trace/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'adc_scan_q8' in r['Kernel_Name']]
start=idx[-9]; t0=int(rows[start]['Start_Timestamp'])
for r in rows[start-2: idx[-4]+2]:
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%-40s q%-3s start %9.1f end %9.1f dur %7.1f' % (r['Kernel_Name'][:40], r.get('Queue_Id','?'), (st-t0)/1e3, (en-t0)/1e3, (en-st)/1e3))
PY
