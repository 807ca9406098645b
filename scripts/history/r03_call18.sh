#!/usr/bin/env bash
# PMC passes of the M=64 byte-table kernel (config 4's shape: 10M x 64 codes, 256 queries)
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c18; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCP_[A-Z_0-9a-z]*\|TA_[A-Z_0-9a-z]*\|TCC_EA0[A-Z_0-9a-z]*\|TCC_REQ[A-Z_a-z0-9]*\|TCC_READ[A-Z_a-z0-9]*" | sort -u > $OUT/counter_names.txt
P="python scripts/prof_scan.py --data lowrank --fused --valid --m 64 --batch 256"
tag=c4_10m
for pass in "a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "c FETCH_SIZE GRBM_GUI_ACTIVE" "d WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "e TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr" "f SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM TCP_TCC_READ_REQ_LATENCY_sum"; do
  set -- $pass; p=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $ROOT/$OUT/${tag}/pmc_$p -- $P --rows 10000000 --iters 4 > $OUT/${tag}_pmc_$p.log 2>&1
  f=$(find $OUT/$tag/pmc_$p -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && (head -1 $f; grep annlite $f) > $OUT/scan_${tag}_q8_pmc_$p.csv || tail -5 $OUT/${tag}_pmc_$p.log
done
python scripts/summarize_prof.py $OUT/$tag > $OUT/scan_${tag}_q8_summary.txt 2>&1
ANNLITE_DEBUG_COUNTERS=2 $P --rows 10000000 --iters 8 > $OUT/scan_${tag}_q8_timeline.txt 2>&1
ANNLITE_DEBUG_COUNTERS=1 $P --rows 10000000 --iters 8 > $OUT/scan_${tag}_q8_debug_counters.txt 2>&1
find $OUT -name '*.db' -delete; find $OUT -type d -name 'pmc_*' -prune -exec rm -rf {} \; 2>/dev/null
tail -40 $OUT/scan_${tag}_q8_summary.txt; grep -h "scan kernel\|timeline\|items:\|byte-table kernel:" $OUT/scan_*_q8_*.txt | cut -c1-330
python - <<'PY'
import csv,glob
for f in sorted(glob.glob('gpurun_out/r03c18/scan_c4_10m_q8_pmc_[ef].csv')):
    acc={}
    for r in csv.DictReader(open(f)):
        if 'adc_scan_q8' in r['Kernel_Name']:
            acc.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
    for k,v in acc.items(): print(f[-10:], k, 'n', len(v), 'mean %.4g' % (sum(v)/len(v)))
PY
