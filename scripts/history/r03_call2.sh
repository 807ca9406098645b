#!/usr/bin/env bash
# round 3, GPU call 2: where the shard-size launch spends its time (phase stamps, PMC), env sweeps
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD
OUT=gpurun_out/r03c2; mkdir -p $OUT
export TMPDIR=/tmp
P="python scripts/prof_scan.py --data lowrank --fused --valid --iters 24"
$P --rows 1250000 > $OUT/plain_1p25m.txt 2>&1
ANNLITE_DEBUG_COUNTERS=1 $P --rows 1250000 > $OUT/timeline_1p25m.txt 2>&1
ANNLITE_DEBUG_COUNTERS=1 ANNLITE_DEBUG_SKIP=4 $P --rows 1250000 > $OUT/timeline_1p25m_skip4.txt 2>&1
ANNLITE_DEBUG_SKIP=4 $P --rows 1250000 > $OUT/plain_1p25m_skip4.txt 2>&1
ANNLITE_DEBUG_COUNTERS=1 $P --rows 10000000 --iters 12 > $OUT/timeline_10m.txt 2>&1
for v in "ANNLITE_SEED_ROWS=16384" "ANNLITE_SEED_ROWS=65536" "ANNLITE_Q8_TUNE=100000,16,384,3" "ANNLITE_Q8_TUNE=15,16,384,1" "ANNLITE_Q8_TUNE=15,16,384,7" "ANNLITE_Q8_TARGET=120" "ANNLITE_SCAN_SLICES=16"; do
  env $v $P --rows 1250000 > "$OUT/sweep_${v//[=,]/_}.txt" 2>&1
done
grep -H "scan kernel\|timeline\|byte-table kernel:" $OUT/*.txt | cut -c1-420
Q="--no-rerank --ivf-cells 0 --cpu-queries 0 --recall-queries 0 --steps 80 --warmup 10 --rows 1250000"
for v in "X=1" "ANNLITE_SEED_ROWS=16384" "ANNLITE_SEED_ROWS=65536" "ANNLITE_SCAN_SLICES=16"; do
  env $v python bench.py $Q --streams 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'ms/step %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done
# PMC at shard size (own runs, kernel-trace only)
for pass in "a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "c FETCH_SIZE GRBM_GUI_ACTIVE" "d WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  set -- $pass; tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $ROOT/$OUT/pmc_$tag -- $P --rows 1250000 --iters 6 > $OUT/pmc_$tag.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -- $P --rows 1250000 --iters 24 > $OUT/trace.log 2>&1
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; tail -40 $OUT/summary.txt
# keep the counter files small: only our kernels' rows
for f in $(find $OUT -name '*counter_collection.csv'); do (head -1 $f; grep annlite $f) > $f.tmp; mv $f.tmp $f; done
find $OUT -name '*.db' -delete; find $OUT -name '*agent_info.csv' -delete
du -sh $OUT
