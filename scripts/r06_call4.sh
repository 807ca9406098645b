#!/usr/bin/env bash
# Round 6, call 4: step-loop microbenchmark with permute addressing (one v_perm_b32 per pair of look-ups), host cost after the
# stream look-up trims, the example's M = 128 shape, forced-exchange shard.
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=gpurun_out/r06c4; mkdir -p $OUT; export TMPDIR=/tmp
for b in a1 a3 a0 a3f2 a1 a3; do ./scripts/step_loop_$b.bin 2000 256; done 2>&1 | tee $OUT/ubench_step_loop_perm.txt
timeout 300 python scripts/host_overhead.py > $OUT/host_overhead.txt 2>&1; grep -E "host us|function calls" $OUT/host_overhead.txt
for shape in "--m 128 --dsub 1" "--m 64 --dsub 2"; do
  tag=$(echo $shape | tr -d ' -')
  timeout 200 python scripts/prof_scan.py --rows 1000000 --batch 256 --data lowrank --fused --valid --iters 6 --layout 0 $shape 2>&1 | grep -v "^/opt" | tail -3 > $OUT/scan_1m_${tag}_plain.txt
  echo "== $shape"; cat $OUT/scan_1m_${tag}_plain.txt
done
A="--legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 200 --warmup 20"
ANNLITE_FORCE_GATHER=1 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --rows 1250000 $A --streams 2 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('shard forced gather: ms/step %.4f host_enqueue %.4f exchange_ms %s sha %s' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d.get('exchange_ms'), d['result_sha256'][:8]))" | tee $OUT/shard_forced_gather.txt
