#!/usr/bin/env bash
# Round 6, call 18: awkward insertion orders for the GPU graph build; bench.py's graph leg with the longer list beside it.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c18; mkdir -p $OUT
timeout 900 python -m pytest tests/test_graph_gpu_build.py -x -q -m gpu -k awkward 2>&1 | tail -8 | tee $OUT/pytest.txt
timeout 900 python bench.py --legs graph --cpu-queries 0 --steps 40 --warmup 5 > $OUT/bench_graph_leg.json 2> $OUT/bench_graph_leg.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06c18/bench_graph_leg.json') if l.startswith('{')][-1])
print('main %.0f q/s %.4f ms' % (d['value'], d['ms_per_step']))
print('graph', d['graph'])
PY
