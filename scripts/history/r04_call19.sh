#!/usr/bin/env bash
# round 4, call 19: HBM traffic of config 5's graph walk (graph_beam_search_kernel, 5M rows, 1024 queries, ef_search 128).  The graph
# is built and dumped by an un-profiled process; rocprofv3 only sees the process that loads it and walks (counters of the walk kernel
# only, two passes, each under its own timeout).
set -u
cd "$(dirname "$0")/../.."; ROOT=$PWD; OUT=gpurun_out/r04c19; mkdir -p $OUT; export TMPDIR=/tmp
G=/tmp/g5m
# (a 20k-row dry run of both halves first: a mistake in the script must not cost the minute the real build takes)
timeout 60 python scripts/prof_graph_walk.py --build /tmp/gsmall --rows 20000 > $OUT/dry_build.log 2>&1 && timeout 40 python scripts/prof_graph_walk.py --walk /tmp/gsmall --rows 20000 --iters 1 > $OUT/dry_walk.log 2>&1 || { echo "dry run failed"; tail -5 $OUT/dry_build.log $OUT/dry_walk.log; exit 1; }
tail -1 $OUT/dry_walk.log
timeout 150 python scripts/prof_graph_walk.py --build $G --rows 5000000 > $OUT/build.log 2>&1; echo "build rc=$?"; tail -2 $OUT/build.log
timeout 70 rocprofv3 --kernel-trace --kernel-include-regex "graph_beam" --pmc FETCH_SIZE GRBM_GUI_ACTIVE -f csv -d $ROOT/$OUT/pmc_c -- python scripts/prof_graph_walk.py --walk $G --rows 5000000 > $OUT/walk_pmc_c.log 2>&1; echo "pmc_c rc=$?"
timeout 70 rocprofv3 --kernel-trace --kernel-include-regex "graph_beam" --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $ROOT/$OUT/pmc_d -- python scripts/prof_graph_walk.py --walk $G --rows 5000000 > $OUT/walk_pmc_d.log 2>&1; echo "pmc_d rc=$?"
find gpurun_out -name "*.db" -delete 2>/dev/null; find gpurun_out -name "*kernel_trace.csv" -delete 2>/dev/null; find gpurun_out -name "*agent_info.csv" -delete 2>/dev/null
python - <<'PY'
import csv, glob, collections
for tag in ('pmc_c', 'pmc_d'):
    acc = collections.defaultdict(list)
    for f in glob.glob('gpurun_out/r04c19/%s/**/*counter_collection.csv' % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'graph_beam' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(tag, {k: (sum(v) / len(v), len(v)) for k, v in acc.items()})
PY
tail -1 $OUT/walk_pmc_c.log; du -sh gpurun_out
