#!/usr/bin/env bash
# Round 6, call 71: the shipped re-rank defaults (rank 1, nearest 2 cells in 4 parts, whole-cell seed) -- rocprofv3 kernel stats at 16 probed
# cells, and the pool sweep at 8 and 32 probed cells.
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=gpurun_out/r06c71; mkdir -p $OUT; export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/rr_trace -- python scripts/prof_ivf_bytes.py --probe 16 --loop 40 --rerank 1 > $OUT/rr_trace.log 2>&1
python - <<PY | tee $OUT/ivf_rerank_kernel_stats_default.txt
import csv,glob
print('command: rocprofv3 --kernel-trace --stats -- python scripts/prof_ivf_bytes.py --probe 16 --loop 40 --rerank 1   (10M x 128, M = 16, 256 cells, 16 probed, 1024 queries, limit 10, rerank_k 16; the index defaults: bound_rank 1, nearest 2 cells in 4 parts, first bound from the whole nearest cell; 3 warm-up + 40 searches)')
for f in glob.glob('$OUT/rr_trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 40 <= int(r['Calls']) <= 50: print('%-92s calls=%-4s avg_us=%8.1f min_us=%8.1f max_us=%8.1f' % (r['Name'][:92], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
grep "^path" $OUT/rr_trace.log | tee -a $OUT/ivf_rerank_kernel_stats_default.txt
rm -rf $OUT/rr_trace
for P in 8 32; do
  echo "n_probe $P" | tee -a $OUT/ivf_rerank_final_sweep_probes.txt
  timeout 100 python scripts/sweep_ivf_rerank.py --probe $P --configs 1:2x4,2:2x4,2:4x4 2>&1 | grep "^{\|Error\|error" | tee -a $OUT/ivf_rerank_final_sweep_probes.txt
done
