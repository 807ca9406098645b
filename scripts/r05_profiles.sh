#!/usr/bin/env bash
# round 5 evidence (profiles/r05/): the default bench line (all legs), rocprofv3 kernel stats of the bench command, PMC passes (each its
# own run, kernel-trace only) of the headline scan, the k = 50 scan (64-key lists) and the packed graph walk, the shard pair behind the
# 8-GPU estimate.   usage: scripts/r05_profiles.sh [part ...]
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=gpurun_out/r05p; mkdir -p $OUT; export TMPDIR=/tmp
PARTS=${@:-bench stats pmc10m pmck50 graph shards}
for part in $PARTS; do case $part in
bench)
  timeout 900 python bench.py > $OUT/bench_10m_n1.json 2>$OUT/bench_10m_n1.err;;
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
  bash scripts/gpu_profile.sh r05_10m --rows 10000000 --data lowrank --fused --valid --iters 6 > $OUT/scan_10m_q8_summary.txt 2>&1;;
pmck50)
  # (the library's own choice for k = 50: the byte-table kernel with 64-key lists once its first guarded launches have completed)
  bash scripts/gpu_profile.sh r05_10m_k50 --rows 10000000 --data lowrank --fused --valid --iters 8 --k 50 > $OUT/scan_10m_k50_q8_summary.txt 2>&1
  ANNLITE_DEBUG_COUNTERS=1 timeout 300 python scripts/prof_scan.py --rows 10000000 --data lowrank --fused --valid --iters 12 --k 50 > $OUT/scan_10m_k50_q8_debug_counters.txt 2>&1;;
graph)
  timeout 600 python scripts/prof_graph_walk.py --build /tmp/g5m --rows 5000000 > $OUT/graph_build.log 2>&1
  for lay in packed plain; do
    rocprofv3 --kernel-trace --kernel-include-regex graph_beam --pmc FETCH_SIZE GRBM_GUI_ACTIVE -f csv -d $ROOT/$OUT/graph_${lay}_c -- python scripts/prof_graph_walk.py --walk /tmp/g5m --rows 5000000 --layout $lay > $OUT/graph_walk_5m_${lay}_c.log 2>&1
    rocprofv3 --kernel-trace --kernel-include-regex graph_beam --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $ROOT/$OUT/graph_${lay}_d -- python scripts/prof_graph_walk.py --walk /tmp/g5m --rows 5000000 --layout $lay > $OUT/graph_walk_5m_${lay}_d.log 2>&1
  done
  python - <<PY > $OUT/graph_walk_5m_pmc_summary.txt
import csv,glob,collections
print('command: scripts/r05_profiles.sh graph (graph built + dumped by an un-profiled process; rocprofv3 --kernel-trace --kernel-include-regex graph_beam --pmc ... -- python scripts/prof_graph_walk.py --walk /tmp/g5m --rows 5000000 --layout L; two passes per layout)')
print(open('$OUT/graph_build.log').read().strip()[-300:])
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
shards)
  for rows in 10000000 1250000; do
    A="--rows $rows --legs none --cpu-queries 0 --recall-queries 0 --no-rerank --steps 200 --warmup 20"
    for s in 1 2; do timeout 300 python bench.py $A --streams $s > $OUT/bench_shard_${rows}_s${s}_200steps.json 2>/dev/null; done
    ANNLITE_FORCE_GATHER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 $A --streams 2 > $OUT/bench_shard_${rows}_forced_gather_200steps.json 2>/dev/null
  done
  python - <<'PY' | tee gpurun_out/r05p/shard_table.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r05p/bench_shard_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
        print('%-52s ms/step %.4f  kernel_ms %.4f frac %.3f clock %.0f MHz q/s %.0f streams %s exchange_ms %s sha %s' % (f.split('/')[-1], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('shader_clock_mhz') or 0, d['value'], d['config'].get('streams'), d.get('exchange_ms'), d['result_sha256'][:12]))
    except Exception as e: print(f, 'ERR', e)
PY
  ;;
esac; find gpurun_out -name '*.db' -delete 2>/dev/null; find gpurun_out -name '*kernel_trace.csv' -delete 2>/dev/null; done
prune() {
  find gpurun_out -name "*.db" -delete 2>/dev/null
  find gpurun_out -name "*kernel_trace.csv" -delete 2>/dev/null
  find gpurun_out -name "*agent_info.csv" -delete 2>/dev/null
  for f in $(find gpurun_out -name '*counter_collection.csv'); do (head -1 $f; grep -E "annlite|graph_beam" $f) > $f.tmp; mv $f.tmp $f; done
  true
}
prune
du -sh gpurun_out
