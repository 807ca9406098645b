#!/usr/bin/env bash
# Round 5, GPU call 9: deterministic training -- the result digest of the same workload must repeat across processes, streams and the forced exchange.
set -u
cd "$(dirname "$0")/../.."; OUT=gpurun_out/r05c9; mkdir -p $OUT
timeout 200 python -m pytest tests/test_round4_gpu.py -x -q -m gpu -k "deterministic_fit" > $OUT/pytest_det.txt 2>&1; echo "det rc=$?"; tail -3 $OUT/pytest_det.txt
A="--rows 1250000 --legs none --cpu-queries 16 --recall-queries 0 --no-rerank --steps 40 --warmup 8"
timeout 200 python bench.py $A --streams 1 > $OUT/digest_run1_s1.json 2>/dev/null
timeout 200 python bench.py $A --streams 2 > $OUT/digest_run2_s2.json 2>/dev/null
ANNLITE_FORCE_GATHER=1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 $A --streams 2 > $OUT/digest_run3_forced_gather.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05c9/digest_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        c=d['cpu_baseline']
        print('%-40s sha %s  codes %s  ms/step %.4f  oracle-equal %s (%s queries)' % (f.split('/')[-1], d['result_sha256'][:16], d['shard_codes_checksum_sum'], d['ms_per_step'], c and c['gpu_matches_cpu_bit_exact_all'], c and c['queries_checked']))
    except Exception as e: print(f,'ERR',e)
PY
