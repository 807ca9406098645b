#!/usr/bin/env bash
# round 4, GPU call 4: full suite on the interleaved M = 64 layout + early merger + slice-per-XCD map; config 4 and shard timings
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r04c4; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.txt
tail -40 $OUT/pytest.txt
C4="--rows 10000000 --m 64 --dsub 12 --batch 256 --iters 15"
timeout 600 python scripts/ab_scan.py $C4 --envs "ANNLITE_Q8_MAP=0|ANNLITE_Q8_MAP=1|ANNLITE_Q8_MAP=1,ANNLITE_SCAN_SLICES=16|ANNLITE_Q8_MAP=1,ANNLITE_SCAN_SLICES=24" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c4.txt
timeout 600 python bench.py --rows 10000000 --dim 768 --m 64 --batch 256 --metric cosine --steps 20 --warmup 5 --cpu-queries 16 --cpu-repeats 1 --recall-queries 32 --legs none > $OUT/bench_c4.json 2>$OUT/err_c4.txt
A="--rows 1250000 --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 200 --warmup 20"
timeout 200 python bench.py $A --streams 2 > $OUT/shard_s2.json 2>$OUT/err_s2.txt
ANNLITE_NO_EARLY_MERGE=1 timeout 200 python bench.py $A --streams 2 > $OUT/shard_s2_noearly.json 2>$OUT/err_s2b.txt
timeout 200 python bench.py $A --streams 2 > $OUT/shard_s2_again.json 2>$OUT/err_s2c.txt
ANNLITE_FORCE_GATHER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 $A --streams 2 > $OUT/shard_gather_s2.json 2>$OUT/err_g2.txt
timeout 300 python bench.py --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 100 --warmup 20 > $OUT/bench_10m_s1.json 2>$OUT/err_10m.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04c4/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f q/s %.0f kernel_ms %.4f frac %.3f exch %s parity %s' % (d['ms_per_step'], d['value'], r['kernel_ms'], r['frac'], d.get('exchange_ms'), (d.get('cpu_baseline') or {}).get('gpu_matches_cpu_bit_exact_all')))
    except Exception as e: print(f, 'ERR', e)
PY
