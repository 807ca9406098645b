#!/usr/bin/env bash
# MFMA evidence for the LUT GEMM: kernel stats + MFMA counters (own pass, kernel-trace only)
OUT=gpurun_out/prof_lut; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.." ; ROOT=$PWD
python scripts/prof_lut.py > $OUT/timing.txt 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -- python scripts/prof_lut.py --iters 5 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -f csv -d $ROOT/$OUT/pmc_a -- python scripts/prof_lut.py --iters 5 > $OUT/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA -f csv -d $ROOT/$OUT/pmc_b -- python scripts/prof_lut.py --iters 5 > $OUT/pmc_b.log 2>&1
python - <<PY > $OUT/summary.txt
import csv,glob,collections
print(open('$OUT/timing.txt').read())
print('== kernel stats ==')
for f in glob.glob('$OUT/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'lut_' in r['Name']: print('%-80s calls=%-4s avg_us=%8.1f' % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3))
for t in ('pmc_a','pmc_b'):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv'%t, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'lut_' in r['Kernel_Name']: acc[(r['Kernel_Name'][:60], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    print('== %s (per-dispatch mean) ==' % t)
    for k,c in acc.items():
        print(' ', k)
        for n,v in sorted(c.items()): print('      %-30s %.4g (n=%d)' % (n, sum(v)/len(v), len(v)))
PY
cat $OUT/summary.txt; tail -3 $OUT/pmc_b.log
