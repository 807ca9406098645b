#!/usr/bin/env bash
# quick PMC look at the scan kernel for one variant: scripts/gpu_pmc_quick.sh <variant> <rows>
V=${1:-0}; ROWS=${2:-10000000}
OUT=gpurun_out/pmcq_v${V}; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=$PWD
ANNLITE_SCAN_VARIANT=$V rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -f csv -d $ROOT/$OUT/a -- python scripts/prof_scan.py --rows $ROWS --iters 2 > $OUT/a.log 2>&1
ANNLITE_SCAN_VARIANT=$V rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE -f csv -d $ROOT/$OUT/b -- python scripts/prof_scan.py --rows $ROWS --iters 2 > $OUT/b.log 2>&1
python - <<PY
import csv,glob,collections
for t in 'ab':
    acc=collections.defaultdict(list)
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv'%t, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'adc_scan' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for c,v in sorted(acc.items()): print('v$V %-24s %.4g'%(c,sum(v)/len(v)))
PY
