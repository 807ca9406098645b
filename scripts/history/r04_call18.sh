#!/usr/bin/env bash
# round 4, GPU call 18: config 5 -- HBM traffic of the GPU graph walk (counters of graph_beam_search_kernel only); the torchrun bench tests
# once more (gather_and_merge now issues one collective)
set -u
cd "$(dirname "$0")/../.."; ROOT=$PWD; OUT=gpurun_out/r04w; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "torchrun or seed_exchange" > $OUT/pytest_torchrun.txt 2>&1; tail -3 $OUT/pytest_torchrun.txt
timeout 400 rocprofv3 --kernel-trace --kernel-include-regex "graph_beam" --pmc FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -f csv -d $ROOT/$OUT/graph -- python scripts/bench_hnsw.py --rows 5000000 --steps 3 > $OUT/config5_5m_under_pmc.json 2>$OUT/graph.log
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
for f in $(find $OUT -name '*counter_collection.csv'); do (head -1 $f; grep annlite $f) > $f.tmp; mv $f.tmp $f; done
python - <<'PY' | tee gpurun_out/r04w/graph_pmc.txt
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('gpurun_out/r04w/graph/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'graph_beam' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('graph_beam_search_kernel, 5M rows x 1024 queries, ef_search 128: per-dispatch mean')
for n,v in sorted(acc.items()): print('   %-20s %.5g (n=%d)' % (n, sum(v)/len(v), len(v)))
PY
tail -c 600 $OUT/config5_5m_under_pmc.json
