#!/usr/bin/env bash
# Profile the hot path on the GPU box: kernel trace + stats, then PMC passes (each in its own run,
# only with --kernel-trace, as the pool requires).  Usage: scripts/gpu_profile.sh <tag> [prof_scan args]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.." ; ROOT=$PWD
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -- python scripts/prof_scan.py "$@" > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -f csv -d $ROOT/$OUT/pmc_a -- python scripts/prof_scan.py "$@" > $OUT/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $ROOT/$OUT/pmc_b -- python scripts/prof_scan.py "$@" > $OUT/pmc_b.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -f csv -d $ROOT/$OUT/pmc_c -- python scripts/prof_scan.py "$@" > $OUT/pmc_c.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $ROOT/$OUT/pmc_d -- python scripts/prof_scan.py "$@" > $OUT/pmc_d.log 2>&1
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
