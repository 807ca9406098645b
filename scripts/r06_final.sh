#!/usr/bin/env bash
# Round 6, final evidence of the tree (third session): GPU suite, the default bench line + rocprofv3 kernel stats, and the pruned search over
# cells: A/B against the u16 pipeline, kernel stats, HBM traffic of the cell-tile scan (two PMC passes, own runs).
#   usage: scripts/r06_final.sh [part ...]   parts: suite bench ivf
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=gpurun_out/r06f; mkdir -p $OUT; export TMPDIR=/tmp
PARTS=${@:-suite bench ivf}
for part in $PARTS; do case $part in
suite)
  timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -6 | tee $OUT/pytest_gpu_suite.txt;;
bench)
  bash scripts/r06_profiles.sh bench stats 2>&1 | grep -v "at::native\|rocprim\|rocclr\|Cijk" | tail -14
  cp gpurun_out/r06p/bench_10m_n1.json gpurun_out/r06p/bench_10m_n1_under_rocprof.json gpurun_out/r06p/bench_10m_n1_rocprof_kernel_stats.txt $OUT/ 2>/dev/null
  rm -rf gpurun_out/r06p/bench_trace;;
ivf)
  timeout 600 python scripts/bench_ivf_bytes.py --probes 8,16,32 --reps 30 2>&1 | grep "n_probe\|exhaustive" | tee $OUT/ivf_bytes_10m.txt
  rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/ivf_trace -- python scripts/prof_ivf_bytes.py --probe 16 --loop 50 > $OUT/ivf_trace.log 2>&1
  python - <<PY | tee $OUT/ivf_kernel_stats_p16.txt
import csv,glob
print('command: rocprofv3 --kernel-trace --stats -- python scripts/prof_ivf_bytes.py --probe 16 --loop 50   (10M x 128, M = 16, 256 cells, 16 probed, 1024 queries, k = 10; 3 warm-up + 50 searches)')
for f in glob.glob('$OUT/ivf_trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 50 <= int(r['Calls']) <= 60: print('%-92s calls=%-4s avg_us=%8.1f min_us=%8.1f max_us=%8.1f' % (r['Name'][:92], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
  rm -rf $OUT/ivf_trace
  rocprofv3 --kernel-trace --kernel-include-regex "adc_scan_q8" --pmc FETCH_SIZE GRBM_GUI_ACTIVE -f csv -d $ROOT/$OUT/ivf_pmc_c -- python scripts/prof_ivf_bytes.py --probe 16 --loop 20 > $OUT/ivf_pmc_c.log 2>&1
  rocprofv3 --kernel-trace --kernel-include-regex "adc_scan_q8" --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $ROOT/$OUT/ivf_pmc_d -- python scripts/prof_ivf_bytes.py --probe 16 --loop 20 > $OUT/ivf_pmc_d.log 2>&1
  python - <<PY | tee $OUT/ivf_scan_pmc_summary.txt
import csv,glob,collections
print('command: rocprofv3 --kernel-trace --kernel-include-regex adc_scan_q8 --pmc {FETCH_SIZE GRBM_GUI_ACTIVE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum} -- python scripts/prof_ivf_bytes.py --probe 16 --loop 20   (two passes, own runs)')
acc=collections.defaultdict(list)
for t in 'cd':
    for f in glob.glob('$OUT/ivf_pmc_%s/**/*counter_collection.csv' % t, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'adc_scan_q8' in r['Kernel_Name'] and 'Lb1ELi16ELb1' in r['Kernel_Name'].replace('true','Lb1').replace(' ','') or ('adc_scan_q8' in r['Kernel_Name'] and r['Kernel_Name'].rstrip(')').rstrip().endswith('true>(annlite::ScanArgs')):
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
if not acc:
    for t in 'cd':
        for f in glob.glob('$OUT/ivf_pmc_%s/**/*counter_collection.csv' % t, recursive=True):
            for r in csv.DictReader(open(f)):
                if 'adc_scan_q8' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
m={c: sum(v)/len(v) for c,v in acc.items()}
print('cell-tile scan, per-dispatch means over %d dispatches: %s' % (len(acc.get('FETCH_SIZE', [])), {c: round(v, 1) for c, v in m.items()}))
if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
    print('HBM bytes per launch = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 = %.4g (gfx950 correction as profiles/traffic.json); L2 hit rate %.1f %%' % (m['FETCH_SIZE']*2048 + m['WRITE_SIZE']*1024, 100*m.get('TCC_HIT_sum',0)/max(1.0, m.get('TCC_HIT_sum',0)+m.get('TCC_MISS_sum',0))))
PY
  rm -rf $OUT/ivf_pmc_c $OUT/ivf_pmc_d;;
esac; done
