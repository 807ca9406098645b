#!/usr/bin/env bash
# round 4, GPU call 3: new tests (early merger), early merger A/B at the shard and at 10M rows, config-4 XCD map A/B + the
# interleaved / no-row-load timing experiments under both maps
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r04c3; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_round4_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "early_merger or multi_gpu or rerank" > $OUT/pytest_new.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_new.txt
tail -15 $OUT/pytest_new.txt
timeout 300 python scripts/ab_scan.py --rows 1250000 --envs "|ANNLITE_NO_EARLY_MERGE=1|ANNLITE_NO_EARLY_MERGE=1,ANNLITE_NO_PREBUILT_TABLES=1|" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_shard.txt
timeout 300 python scripts/ab_scan.py --rows 10000000 --iters 20 --envs "|ANNLITE_NO_EARLY_MERGE=1|ANNLITE_NO_EARLY_MERGE=1,ANNLITE_NO_PREBUILT_TABLES=1|" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_10m.txt
C4="--rows 10000000 --m 64 --dsub 12 --batch 256 --iters 15"
timeout 600 python scripts/ab_scan.py $C4 --envs "ANNLITE_Q8_MAP=0|ANNLITE_Q8_MAP=1|ANNLITE_Q8_MAP=0,ANNLITE_NO_EARLY_MERGE=1|ANNLITE_Q8_MAP=1,ANNLITE_NO_EARLY_MERGE=1" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c4.txt
for v in q8exp4 q8exp3; do
  ANNLITE_HIP_LIB=build_exp/lib_$v.so timeout 600 python scripts/ab_scan.py $C4 --envs "ANNLITE_Q8_MAP=0|ANNLITE_Q8_MAP=1" 2>&1 | grep -v amdgpu.ids | sed "s/^/$v: /" | tee -a $OUT/ab_c4.txt
done
