#!/usr/bin/env bash
# kernel timeline (durations + gaps) of the tail of a command: scripts/gpu_timeline.sh <tag> <n_last> -- <cmd...>
TAG=$1; NLAST=$2; shift 3
OUT=gpurun_out/tl_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.." ; ROOT=$PWD
rocprofv3 --kernel-trace -f csv -d $ROOT/$OUT/trace -- "$@" > $OUT/trace.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('$OUT/trace/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
prev=None
for r in rows[-$NLAST:]:
    st,en=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(st-prev)/1e3 if prev else 0
    print('%-50s dur %8.1f us  gap_before %7.1f us' % (r['Kernel_Name'][:50], (en-st)/1e3, gap))
    prev=en
PY
