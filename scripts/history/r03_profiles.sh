#!/usr/bin/env bash
# round 3: the profiles committed under profiles/r03/ -- rocprofv3 kernel stats of the bench command, PMC passes of the
# byte-table kernel at the headline size and at one rank's share of it (each pass its own run, kernel-trace only)
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03prof; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
P="python scripts/prof_scan.py --data lowrank --fused --valid"
for rows in 1250000 10000000; do
  tag=$([ $rows = 1250000 ] && echo 1p25m || echo 10m)
  it=$([ $rows = 1250000 ] && echo 8 || echo 5)
  for pass in "a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "c FETCH_SIZE GRBM_GUI_ACTIVE" "d WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    set -- $pass; p=$1; shift
    timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $ROOT/$OUT/${tag}/pmc_$p -- $P --rows $rows --iters $it > $OUT/${tag}_pmc_$p.log 2>&1
  done
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/${tag}/trace -- $P --rows $rows --iters 24 > $OUT/${tag}/trace.log 2>&1
  python scripts/summarize_prof.py $OUT/$tag > $OUT/scan_${tag}_q8_summary.txt 2>&1
  ANNLITE_DEBUG_COUNTERS=1 $P --rows $rows --iters 12 > $OUT/scan_${tag}_q8_debug_counters.txt 2>&1
  ANNLITE_DEBUG_COUNTERS=2 $P --rows $rows --iters 12 > $OUT/scan_${tag}_q8_timeline.txt 2>&1
  ANNLITE_DEBUG_COUNTERS=2 ANNLITE_DEBUG_SKIP=4 $P --rows $rows --iters 12 > $OUT/scan_${tag}_q8_timeline_no_candidates.txt 2>&1
  for p in a b c d; do
    f=$(find $OUT/$tag/pmc_$p -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && (head -1 $f; grep annlite $f) > $OUT/scan_${tag}_q8_pmc_$p.csv
  done
done
# the bench command itself under the kernel trace (legs off: the trace is of the timed path)
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/bench_trace -- python bench.py --legs none --cpu-queries 0 --recall-queries 0 > $OUT/bench_10m_n1_under_rocprof.json 2> $OUT/bench_trace.log
f=$(find $OUT/bench_trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/bench_10m_n1_rocprof_kernel_stats.csv
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r03prof/bench_10m_n1_rocprof_kernel_stats.csv'):
    with open(f.replace('.csv', '.txt'), 'w') as o:
        for r in csv.DictReader(open(f)):
            o.write('%-100s calls=%-5s avg_ns=%-12s min_ns=%-10s max_ns=%-10s total_ns=%-14s pct=%s\n' % (r['Name'][:100], r['Calls'], r['AverageNs'], r.get('MinNs'), r.get('MaxNs'), r['TotalDurationNs'], r['Percentage']))
    print(open(f.replace('.csv', '.txt')).read()[:1500])
PY
for st in 1 2; do python bench.py --rows 1250000 --legs none --cpu-queries 16 --cpu-repeats 3 --streams $st > $OUT/bench_1p25m_n1_s$st.json 2>/dev/null; done
for rows in 5000000 2500000; do python bench.py --rows $rows --legs none --cpu-queries 0 --recall-queries 0 > $OUT/bench_${rows}_n1.json 2>/dev/null; done
python bench.py --legs none --cpu-queries 0 --recall-queries 0 --streams 2 > $OUT/bench_10m_n1_s2.json 2>/dev/null
find $OUT -name '*.db' -delete; find $OUT -type d -name 'pmc_*' -prune -exec rm -rf {} \; 2>/dev/null; rm -rf $OUT/*/trace $OUT/bench_trace
cat $OUT/scan_1p25m_q8_summary.txt | tail -45; grep -h "scan kernel\|timeline\|items:\|byte-table kernel:" $OUT/scan_*_q8_*.txt | cut -c1-330
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03prof/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.4f  kernel_ms %.4f  frac %.3f  q/s %.0f streams %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], d['config']['streams']))
    except Exception as e: print(f, 'ERR', e)
PY
du -sh $OUT
