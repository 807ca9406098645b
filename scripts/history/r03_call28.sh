#!/usr/bin/env bash
# PMC passes of the M=8 / uint16-code byte-table kernels (Ks = 512: two entry groups; Ks = 768: one)
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c28; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for ks in 512 768; do
  P="python scripts/prof_scan.py --data lowrank --fused --valid --m 8 --dsub 16 --ks $ks --rows 10000000 --iters 4"
  tag=m8_ks${ks}_10m
  $P 2>&1 | grep "scan kernel ms\|kernel choice\|Error\|error" | cut -c1-200
  for pass in "a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "c FETCH_SIZE GRBM_GUI_ACTIVE" "d WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    set -- $pass; p=$1; shift
    timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $ROOT/$OUT/${tag}/pmc_$p -- $P > $OUT/${tag}_pmc_$p.log 2>&1
    f=$(find $OUT/$tag/pmc_$p -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && (head -1 $f; grep adc_scan_q8 $f) > $OUT/scan_${tag}_q8_pmc_$p.csv || tail -3 $OUT/${tag}_pmc_$p.log
  done
done
find $OUT -name '*.db' -delete; find $OUT -type d -name 'pmc_*' -prune -exec rm -rf {} \; 2>/dev/null
python - <<'PY'
import csv,glob
for f in sorted(glob.glob('gpurun_out/r03c28/scan_*_q8_pmc_[a-d].csv')):
    acc={}
    for r in csv.DictReader(open(f)):
        acc.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
    print(f.split('/')[-1], {k:'%.4g'%(sum(v)/len(v)) for k,v in acc.items()})
PY
