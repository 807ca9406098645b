#!/usr/bin/env bash
# Round 6, call 62: the randomised parity runs on the final tree (flat search 100 s, cells + candidate lists 60 s) and rocprofv3 kernel stats
# of the float re-rank on the cell tiles (bound_rank 2 and 1) beside the plain pruned search.
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=gpurun_out/r06c62; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python tests/fuzz_parity.py --seconds 100 --seed 63 2>&1 | tail -4 | tee $OUT/fuzz_parity_seed63.txt
timeout 150 python tests/fuzz_parity.py --cells --seconds 60 --seed 173 2>&1 | tail -4 | tee $OUT/fuzz_parity_cells_seed173.txt
for r in 2 1; do
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/rr_trace_$r -- python scripts/prof_ivf_bytes.py --probe 16 --loop 40 --rerank $r > $OUT/rr_trace_$r.log 2>&1
  python - <<PY | tee $OUT/ivf_rerank_kernel_stats_rank$r.txt
import csv,glob
print('command: rocprofv3 --kernel-trace --stats -- python scripts/prof_ivf_bytes.py --probe 16 --loop 40 --rerank $r   (10M x 128, M = 16, 256 cells, 16 probed, 1024 queries, limit 10, rerank_k 16, bound_rank $r; 3 warm-up + 40 searches)')
for f in glob.glob('$OUT/rr_trace_$r/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 40 <= int(r['Calls']) <= 50: print('%-92s calls=%-4s avg_us=%8.1f min_us=%8.1f max_us=%8.1f' % (r['Name'][:92], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
  rm -rf $OUT/rr_trace_$r
done
