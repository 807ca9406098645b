#!/usr/bin/env bash
# round 4, GPU call 15: small-kernel evidence (LUT MFMA, encode: timing + counters), config 4 traffic / clock under the slice-per-XCD map
# (random codes: the data set does not change the traffic of the row stream); every step under its own timeout
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=gpurun_out/r04t; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python scripts/prof_lut.py > $OUT/lut_timing.txt 2>&1
timeout 90 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/lut_trace -- python scripts/prof_lut.py --iters 5 > $OUT/lut_trace.log 2>&1
timeout 90 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -f csv -d $ROOT/$OUT/lut_a -- python scripts/prof_lut.py --iters 5 > $OUT/lut_a.log 2>&1
timeout 90 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA FETCH_SIZE WRITE_SIZE -f csv -d $ROOT/$OUT/lut_b -- python scripts/prof_lut.py --iters 5 > $OUT/lut_b.log 2>&1
timeout 90 python scripts/bench_encode.py > $OUT/encode.jsonl 2>$OUT/encode.err
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -f csv -d $ROOT/$OUT/enc_a -- python scripts/bench_encode.py > /dev/null 2>$OUT/enc_a.log
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -f csv -d $ROOT/$OUT/enc_b -- python scripts/bench_encode.py > /dev/null 2>$OUT/enc_b.log
C4="--rows 10000000 --m 64 --dsub 12 --batch 256 --data random --fused --valid --iters 4"
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -f csv -d $ROOT/$OUT/c4_c -- python scripts/prof_scan.py $C4 > $OUT/c4_c.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_LDS_IDX_ACTIVE -f csv -d $ROOT/$OUT/c4_d -- python scripts/prof_scan.py $C4 > $OUT/c4_d.log 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
for f in $(find $OUT -name '*counter_collection.csv'); do (head -1 $f; grep annlite $f) > $f.tmp; mv $f.tmp $f; done
python - <<'PY' | tee gpurun_out/r04t/small_kernels_pmc.txt
import csv,glob,collections
print(open('gpurun_out/r04t/lut_timing.txt').read())
for f in glob.glob('gpurun_out/r04t/lut_trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'lut_' in r['Name']: print('%-80s calls=%-4s avg_us=%8.1f' % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3))
for tag,pat in (('lut_a','lut_'),('lut_b','lut_'),('enc_a','encode'),('enc_b','encode'),('c4_c','adc_scan_q8'),('c4_d','adc_scan_q8')):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('gpurun_out/r04t/%s/**/*counter_collection.csv'%tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']: acc[(r['Kernel_Name'][:64], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    print('== %s: per-dispatch mean ==' % tag)
    for k,c in acc.items():
        print(' ', k)
        for n,v in sorted(c.items()): print('      %-28s %.5g (n=%d)' % (n, sum(v)/len(v), len(v)))
print(open('gpurun_out/r04t/encode.jsonl').read())
PY
grep -h "scan kernel ms" $OUT/c4_c.log $OUT/c4_d.log; du -sh gpurun_out
