#!/usr/bin/env bash
# round 4, GPU call 13 (tight budget): PMC passes + phase timeline at the 1.25M-row shard; every step under its own timeout
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=gpurun_out/r04q; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
P="--rows 1250000 --data lowrank --fused --valid --iters 12"
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -f csv -d $ROOT/$OUT/pmc_a -- python scripts/prof_scan.py $P > $OUT/pmc_a.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $ROOT/$OUT/pmc_b -- python scripts/prof_scan.py $P > $OUT/pmc_b.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -f csv -d $ROOT/$OUT/pmc_c -- python scripts/prof_scan.py $P > $OUT/pmc_c.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $ROOT/$OUT/pmc_d -- python scripts/prof_scan.py $P > $OUT/pmc_d.log 2>&1
ANNLITE_DEBUG_COUNTERS=2 timeout 60 python scripts/prof_scan.py --rows 1250000 --data lowrank --fused --valid --iters 30 > $OUT/scan_1p25m_q8_timeline.txt 2>&1
ANNLITE_DEBUG_COUNTERS=1 timeout 60 python scripts/prof_scan.py --rows 1250000 --data lowrank --fused --valid --iters 30 > $OUT/scan_1p25m_q8_debug_counters.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
for f in $(find $OUT -name '*counter_collection.csv'); do (head -1 $f; grep annlite $f) > $f.tmp; mv $f.tmp $f; done
python - <<'PY'
import csv,glob,collections
for t in 'abcd':
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('gpurun_out/r04q/pmc_%s/**/*counter_collection.csv'%t, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'adc_scan_q8' in r['Kernel_Name'] or 'seed_bound' in r['Kernel_Name']: acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,c in acc.items():
        print(t, k)
        for n,v in sorted(c.items()): print('      %-24s %.5g (n=%d)' % (n, sum(v)/len(v), len(v)))
PY
tail -4 $OUT/scan_1p25m_q8_timeline.txt | cut -c1-300; du -sh gpurun_out
