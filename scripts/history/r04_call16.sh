#!/usr/bin/env bash
# round 4, GPU call 16: config 4 (10M x 768-d, m = 64, 256 queries, the bench's data) -- HBM traffic, clock, VALU / LDS activity of the scan
# kernel under the slice-per-XCD map; counters collected for the scan kernel ONLY (--kernel-include-regex), every pass under a timeout
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=gpurun_out/r04u; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
C4="--rows 10000000 --m 64 --dsub 12 --batch 256 --data lowrank --fused --valid --iters 5"
timeout 280 rocprofv3 --kernel-trace --kernel-include-regex "adc_scan_q8" --pmc FETCH_SIZE GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -f csv -d $ROOT/$OUT/c4_c -- python scripts/prof_scan.py $C4 > $OUT/c4_c.log 2>&1
timeout 280 rocprofv3 --kernel-trace --kernel-include-regex "adc_scan_q8" --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -f csv -d $ROOT/$OUT/c4_d -- python scripts/prof_scan.py $C4 > $OUT/c4_d.log 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
for f in $(find $OUT -name '*counter_collection.csv'); do (head -1 $f; grep annlite $f) > $f.tmp; mv $f.tmp $f; done
python - <<'PY' | tee gpurun_out/r04u/c4_pmc.txt
import csv,glob,collections
for tag in ('c4_c','c4_d'):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('gpurun_out/r04u/%s/**/*counter_collection.csv'%tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'adc_scan_q8' in r['Kernel_Name']: acc[(r['Kernel_Name'][:64], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    print('== %s: per-dispatch mean ==' % tag)
    for k,c in acc.items():
        print(' ', k)
        for n,v in sorted(c.items()): print('      %-28s %.5g (n=%d)' % (n, sum(v)/len(v), len(v)))
PY
grep -h "scan kernel ms\|kernel choice" $OUT/c4_c.log $OUT/c4_d.log; du -sh gpurun_out
