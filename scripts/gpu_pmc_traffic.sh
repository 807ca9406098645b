#!/usr/bin/env bash
# HBM traffic of the scan kernel: scripts/gpu_pmc_traffic.sh <tag> [prof_scan args]
TAG=${1:-t}; shift || true
OUT=gpurun_out/traffic_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.." ; ROOT=$PWD
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -f csv -d $ROOT/$OUT/c -- python scripts/prof_scan.py "$@" --iters 3 > $OUT/c.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $ROOT/$OUT/d -- python scripts/prof_scan.py "$@" --iters 3 > $OUT/d.log 2>&1
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for t in 'cd':
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv'%t, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'adc_scan' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for c,v in sorted(acc.items()): print('%-18s %.4g' % (c, sum(v)/len(v)))
if 'FETCH_SIZE' in acc: print('HBM read MB (FETCH_SIZE x 2 x 1024):', sum(acc['FETCH_SIZE'])/len(acc['FETCH_SIZE'])*2048/1e6)
PY
