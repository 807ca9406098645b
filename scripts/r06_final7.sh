#!/usr/bin/env bash
# Round 6, the last evidence of the tree (fifth session): the whole GPU suite, then the bench's rerank / ivf legs (the headline's own line and
# the other legs: final6, one commit earlier -- the scan kernels are the same objects).
set -u
cd "$(dirname "$0")/.."; ROOT=$PWD; OUT=gpurun_out/r06f7; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python -m pytest tests/ -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -25 | tee $OUT/pytest_gpu_suite.txt
timeout 200 python bench.py --legs rerank,ivf --cpu-queries 0 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
for n in ('rerank', 'rerank16', 'rerank16_whole_cells', 'rerank16_rank2', 'rerank16_top16'): print(n, json.dumps(r['ivf'].get(n)))
print(json.dumps(r['summary']))" | tee $OUT/bench_ivf_leg.txt
