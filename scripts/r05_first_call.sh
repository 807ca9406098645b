#!/usr/bin/env bash
# First GPU call of the next round: where things stand on today's box, in ~5 minutes of box time (each step under its own timeout).
#   the GPU suite, the default bench line, the shard pair behind the 8-GPU estimate (10M and 1.25M rows, two streams, exchange forced),
#   the M = 32 shape (byte tables / u16 tables), and k = 50 (u16 tables) -- the baselines of DESIGN section 10.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r05first; mkdir -p $OUT
timeout 260 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "suite rc=$?"; tail -3 $OUT/pytest_gpu.txt
timeout 240 python bench.py > $OUT/bench_10m_n1.json 2>$OUT/bench_10m_n1.err; echo "bench rc=$?"
for rows in 10000000 1250000; do
  A="--rows $rows --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 200 --warmup 20"
  timeout 120 python bench.py $A --streams 2 > $OUT/bench_shard_${rows}_s2.json 2>/dev/null
  ANNLITE_FORCE_GATHER=1 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 $A --streams 2 > $OUT/bench_shard_${rows}_forced_gather.json 2>/dev/null
done
# slice-per-XCD map for M = 16 (what M = 64 got in round 4): A/B on this box, 10M and 1.25M rows, one stream
for rows in 10000000 1250000; do
  for map in 0 1; do
    ANNLITE_Q8_MAP=$map timeout 120 python bench.py --rows $rows --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 100 --warmup 20 > $OUT/bench_map${map}_${rows}.json 2>/dev/null
  done
done
P="--rows 10000000 --data lowrank --fused --valid --iters 8"
ANNLITE_SCAN_VARIANT=50 timeout 60 python scripts/prof_scan.py $P --m 32 --dsub 4 --k 10 2>&1 | grep -v "^/opt" | head -2 > $OUT/scan_10m_m32_q8.txt
ANNLITE_SCAN_VARIANT=31 timeout 60 python scripts/prof_scan.py $P --m 32 --dsub 4 --k 10 2>&1 | grep -v "^/opt" | head -2 > $OUT/scan_10m_m32_u16.txt
timeout 60 python scripts/prof_scan.py $P --k 50 2>&1 | grep -v "^/opt" | head -2 > $OUT/scan_10m_k50_u16.txt
# what 16 row slices cost the byte-table kernel at k = 16 (two work items per workgroup): the shape a 16-key-list k = 50 plan would run
timeout 60 python scripts/prof_scan.py $P --k 16 2>&1 | grep -v "^/opt" | head -2 > $OUT/scan_10m_k16_q8_8slices.txt
ANNLITE_SCAN_SLICES=16 timeout 60 python scripts/prof_scan.py $P --k 16 2>&1 | grep -v "^/opt" | head -2 > $OUT/scan_10m_k16_q8_16slices.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05first/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1]); r = d['roofline']
        print('%-46s q/s %9.0f  ms/step %.4f  kernel_ms %.4f  frac %.3f  streams %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], d['config'].get('streams')))
        for leg in ('c2', 'c4', 'c5', 'm32', 'uniform', 'facade', 'ivf', 'rerank'):
            v = d.get(leg)
            if isinstance(v, dict) and 'value' in v: print('    %-8s %10.0f q/s' % (leg, v['value']))
    except Exception as e:
        print(f, 'ERR', e)
PY
for f in $OUT/scan_10m_*.txt; do echo "$f: $(tail -1 $f | cut -c1-120)"; done
