#!/usr/bin/env bash
# round 6 evidence (profiles/r06/): the default bench line (all legs), rocprofv3 kernel stats of the bench command, PMC passes (each its own
# run, kernel-trace only) of the headline scan + HBM traffic of every shape profiles/traffic.json holds for adc_scan_q8_kernel (its
# revision moved: the M = 16 table image), the shard pair behind the 8-GPU estimate, the small kernels' counters re-taken.
#   usage: scripts/r06_profiles.sh [part ...]
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=gpurun_out/r06p; mkdir -p $OUT; export TMPDIR=/tmp
PARTS=${@:-bench stats pmc10m traffic shards small}
for part in $PARTS; do case $part in
bench)
  timeout 900 python bench.py > $OUT/bench_10m_n1.json 2>$OUT/bench_10m_n1.err; tail -c 1300 $OUT/bench_10m_n1.json; echo;;
stats)
  rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/bench_trace -- python bench.py --legs none --cpu-queries 0 --recall-queries 0 > $OUT/bench_10m_n1_under_rocprof.json 2>$OUT/bench_trace.log
  python - <<PY > $OUT/bench_10m_n1_rocprof_kernel_stats.txt
import csv,glob
print('command: rocprofv3 --kernel-trace --stats -- python bench.py --legs none --cpu-queries 0 --recall-queries 0   (the default workload: 64 set-up + 20 warm-up + 200 timed steps over 4 rotating query batches + the host-transfer and roofline legs)')
print([l for l in open('$OUT/bench_10m_n1_under_rocprof.json') if l.startswith('{')][-1].strip()[:600])
for f in glob.glob('$OUT/bench_trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage']) > 0.005: print('%-86s calls=%-5s avg_us=%9.1f min_us=%9.1f max_us=%9.1f pct=%s' % (r['Name'][:86], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Percentage']))
PY
  cat $OUT/bench_10m_n1_rocprof_kernel_stats.txt;;
pmc10m)
  bash scripts/gpu_profile.sh r06_10m --rows 10000000 --data lowrank --fused --valid --iters 6 > $OUT/scan_10m_q8_summary.txt 2>&1; tail -30 $OUT/scan_10m_q8_summary.txt;;
traffic)
  # FETCH_SIZE / WRITE_SIZE / L2 hits of the scan kernel per shape (two passes each)
  bash scripts/gpu_pmc_traffic.sh r06_1p25m --rows 1250000 --data lowrank --fused --valid > $OUT/traffic_1p25m.txt 2>&1
  bash scripts/gpu_pmc_traffic.sh r06_1m --rows 1000000 --data lowrank --fused --valid > $OUT/traffic_1m.txt 2>&1
  bash scripts/gpu_pmc_traffic.sh r06_10m_k50 --rows 10000000 --data lowrank --fused --valid --k 50 > $OUT/traffic_10m_k50.txt 2>&1
  bash scripts/gpu_pmc_traffic.sh r06_10m_m32 --rows 10000000 --data lowrank --fused --valid --m 32 --dsub 4 > $OUT/traffic_10m_m32.txt 2>&1
  bash scripts/gpu_pmc_traffic.sh r06_c4 --rows 10000000 --m 64 --dsub 12 --batch 256 --data lowrank --fused --valid > $OUT/traffic_c4.txt 2>&1
  for f in $OUT/traffic_*.txt; do echo "== $f"; cat $f; done;;
shards)
  for rows in 10000000 1250000; do
    A="--rows $rows --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 200 --warmup 20"
    for s in 1 2; do timeout 300 python bench.py $A --streams $s > $OUT/bench_shard_${rows}_s${s}_200steps.json 2>/dev/null; done
    ANNLITE_FORCE_GATHER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 $A --streams 2 > $OUT/bench_shard_${rows}_forced_gather_200steps.json 2>/dev/null
  done
  python - <<'PY' | tee gpurun_out/r06p/shard_table.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r06p/bench_shard_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
        print('%-52s ms/step %.4f  kernel_ms %.4f frac %.3f clock %.0f MHz q/s %.0f streams %s exchange_ms %s host_enqueue %.4f sha %s' % (f.split('/')[-1], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('shader_clock_mhz') or 0, d['value'], d['config'].get('streams'), d.get('exchange_ms'), d['host_enqueue_ms_per_step'], d['result_sha256'][:12]))
    except Exception as e: print(f, 'ERR', e)
PY
  ;;
graph)
  # config 5's walk alone (GPU-built graph, pair walk over packed records; the plain one-at-a-time walk beside it): HBM traffic passes
  timeout 600 python scripts/prof_graph_walk.py --build /tmp/g5m --rows 5000000 > $OUT/graph_build.log 2>&1
  for lay in packed plain; do
    rocprofv3 --kernel-trace --kernel-include-regex graph_beam --pmc FETCH_SIZE GRBM_GUI_ACTIVE -f csv -d $ROOT/$OUT/graph_${lay}_c -- python scripts/prof_graph_walk.py --walk /tmp/g5m --rows 5000000 --layout $lay > $OUT/graph_walk_5m_${lay}_c.log 2>&1
    rocprofv3 --kernel-trace --kernel-include-regex graph_beam --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $ROOT/$OUT/graph_${lay}_d -- python scripts/prof_graph_walk.py --walk /tmp/g5m --rows 5000000 --layout $lay > $OUT/graph_walk_5m_${lay}_d.log 2>&1
  done
  python - <<PY > $OUT/graph_walk_5m_pmc_summary.txt
import csv,glob,collections
print('command: scripts/r06_profiles.sh graph (graph built + dumped by an un-profiled process; rocprofv3 --kernel-trace --kernel-include-regex graph_beam --pmc ... -- python scripts/prof_graph_walk.py --walk /tmp/g5m --rows 5000000 --layout L; two passes per layout)')
print(open('$OUT/graph_build.log').read().strip()[-400:])
for lay in ('packed', 'plain'):
    acc=collections.defaultdict(list)
    for t in 'cd':
        print('%s pass %s: %s' % (lay, t, [l.strip() for l in open('$OUT/graph_walk_5m_%s_%s.log' % (lay, t)) if l.startswith('graph walk')][-1:]))
        for f in glob.glob('$OUT/graph_%s_%s/**/*counter_collection.csv' % (lay, t), recursive=True):
            for r in csv.DictReader(open(f)):
                if 'graph_beam' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    m={c: sum(v)/len(v) for c,v in acc.items()}
    print('  %s: per-dispatch means over %d dispatches: %s' % (lay, len(acc.get('FETCH_SIZE', [])), {c: round(v, 1) for c, v in m.items()}))
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        print('  %s: HBM bytes per launch = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 = %.4g; L2 hit rate %.1f %%' % (lay, m['FETCH_SIZE']*2048 + m['WRITE_SIZE']*1024, 100*m.get('TCC_HIT_sum',0)/max(1.0, m.get('TCC_HIT_sum',0)+m.get('TCC_MISS_sum',0))))
PY
  cat $OUT/graph_walk_5m_pmc_summary.txt;;
small)
  bash scripts/gpu_profile_lut.sh > $OUT/lut_mfma_summary.txt 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -f csv -d $ROOT/$OUT/enc_a -- python scripts/bench_encode.py > $OUT/encode.jsonl 2>$OUT/enc_a.log
  rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -f csv -d $ROOT/$OUT/enc_b -- python scripts/bench_encode.py > /dev/null 2>$OUT/enc_b.log
  python - <<PY > $OUT/small_kernels_pmc.txt
import csv,glob,collections
print(open('$OUT/lut_mfma_summary.txt').read())
print(open('$OUT/encode.jsonl').read())
for tag,pat in (('enc_a','encode'),('enc_b','encode')):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv'%tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']: acc[(r['Kernel_Name'][:70], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    print('== %s: per-dispatch mean ==' % tag)
    for k,c in acc.items():
        print(' ', k)
        for n,v in sorted(c.items()): print('      %-26s %.5g (n=%d)' % (n, sum(v)/len(v), len(v)))
PY
  tail -60 $OUT/small_kernels_pmc.txt;;
esac
  # prune after EVERY part: the box's gpurun_out/ comes back only below 64 MiB (call 7 lost its files to a 55-minute limit with
  # the pruning at the end of the script)
  find gpurun_out -name '*.db' -delete 2>/dev/null; find gpurun_out -name '*kernel_trace.csv' -delete 2>/dev/null
  find gpurun_out -name "*agent_info.csv" -delete 2>/dev/null; find gpurun_out -name "*_domain_stats.csv" -delete 2>/dev/null
  for f in $(find gpurun_out -name '*counter_collection.csv'); do (head -1 $f; grep -E "annlite|graph_beam" $f) > $f.tmp; mv $f.tmp $f; done
  rm -rf gpurun_out/r06p/bench_trace
done
du -sh gpurun_out
