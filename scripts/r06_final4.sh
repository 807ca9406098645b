#!/usr/bin/env bash
# Round 6, final evidence of the tree (fourth session): the cell-tile tests first (the candidate generator's default bound changed),
# the whole GPU suite, the default bench line, rocprofv3 kernel stats of the same command (every rocprofv3 under `timeout`).
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=gpurun_out/r06f4; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ivf_byte_tiles.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -6 | tee $OUT/pytest_ivf.txt
timeout 1200 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -6 | tee $OUT/pytest_gpu_suite.txt
timeout 900 python bench.py > $OUT/bench_10m_n1.json 2>$OUT/bench_10m_n1.err; tail -c 1500 $OUT/bench_10m_n1.json; echo
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/bench_trace -- python bench.py --legs none --cpu-queries 0 --recall-queries 0 > $OUT/bench_10m_n1_under_rocprof.json 2>$OUT/bench_trace.log
python - <<PY > $OUT/bench_10m_n1_rocprof_kernel_stats.txt
import csv,glob
print('command: rocprofv3 --kernel-trace --stats -- python bench.py --legs none --cpu-queries 0 --recall-queries 0   (the default workload: 64 set-up + 20 warm-up + 200 timed steps over 4 rotating query batches + the host-transfer and roofline legs)')
print([l for l in open('$OUT/bench_10m_n1_under_rocprof.json') if l.startswith('{')][-1].strip()[:600])
for f in glob.glob('$OUT/bench_trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage']) > 0.005: print('%-86s calls=%-5s avg_us=%9.1f min_us=%9.1f max_us=%9.1f pct=%s' % (r['Name'][:86], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Percentage']))
PY
cat $OUT/bench_10m_n1_rocprof_kernel_stats.txt
rm -rf $OUT/bench_trace
