#!/usr/bin/env bash
# Round 5, GPU call 40: randomised parity run on the final tree (tests/fuzz_parity.py, two seeds)
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; OUT=gpurun_out/r05c40; mkdir -p $OUT
timeout 200 python tests/fuzz_parity.py --seconds 140 --seed 1 > $OUT/fuzz_seed1.txt 2>&1; echo "seed 1 rc=$?"; tail -3 $OUT/fuzz_seed1.txt
timeout 200 python tests/fuzz_parity.py --seconds 140 --seed 2 > $OUT/fuzz_seed2.txt 2>&1; echo "seed 2 rc=$?"; tail -3 $OUT/fuzz_seed2.txt
