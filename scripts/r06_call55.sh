#!/usr/bin/env bash
# (ANNLITE_IVF_CAND_RANK was this call's switch; it has since become the C entry's bound_rank argument / IvfPQGpuIndex.rerank_bound_rank)
# Round 6, call 55: candidate generator on the cell tiles -- the first bound's rank (1 / 2 / 4 x k) against pool size, recall and rate.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c55; mkdir -p $OUT
for f in 4 2 1; do
  echo "ANNLITE_IVF_CAND_RANK=$f" | tee -a $OUT/ivf_cand_rank.txt
  ANNLITE_IVF_CAND_RANK=$f timeout 600 python bench.py --legs rerank,ivf --cpu-queries 0 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print(json.dumps(r['ivf'].get('rerank16'))); print(json.dumps(r['ivf'].get('rerank')))" | tee -a $OUT/ivf_cand_rank.txt
done
timeout 600 python -m pytest tests/test_ivf_byte_tiles.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -5 | tee $OUT/pytest_ivf.txt
