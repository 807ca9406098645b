#!/usr/bin/env bash
# round 4, GPU call 5: config 4 -- plane layout vs row-major on the same box, candidate handling off, table resolution, kernel timeline
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r04c5; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
C4="--rows 10000000 --m 64 --dsub 12 --batch 256 --iters 15"
E="|ANNLITE_DEBUG_SKIP=4|ANNLITE_Q8_TARGET=256|ANNLITE_Q8_TARGET=512|ANNLITE_Q8_TARGET=768|"
timeout 600 python scripts/ab_scan.py $C4 --envs "$E" 2>&1 | grep -v amdgpu.ids | sed "s/^/planes:   /" | tee $OUT/ab_c4.txt
ANNLITE_HIP_LIB=build_exp/lib_rowmajor64.so timeout 600 python scripts/ab_scan.py $C4 --envs "$E" 2>&1 | grep -v amdgpu.ids | sed "s/^/rowmajor: /" | tee -a $OUT/ab_c4.txt
timeout 600 python scripts/ab_scan.py $C4 --envs "|ANNLITE_DEBUG_SKIP=4" 2>&1 | grep -v amdgpu.ids | sed "s/^/planes:   /" | tee -a $OUT/ab_c4.txt
bash scripts/gpu_timeline.sh c4 14 -- python bench.py --rows 10000000 --dim 768 --m 64 --batch 256 --metric cosine --steps 6 --warmup 2 --prewarm-steps 4 --cpu-queries 0 --recall-queries 0 --no-rerank --legs none > $OUT/timeline_c4.txt 2>&1
tail -16 $OUT/timeline_c4.txt
