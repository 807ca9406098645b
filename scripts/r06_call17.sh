#!/usr/bin/env bash
# Round 6, call 17: bench.py's new `graph` leg (HNSW-over-PQ over the headline's 10M rows, built on the GPU) beside the re-rank leg.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c17; mkdir -p $OUT
timeout 900 python bench.py --legs rerank,graph,ivf --cpu-queries 0 --steps 40 --warmup 5 > $OUT/bench_graph_leg.json 2> $OUT/bench_graph_leg.err
tail -3 $OUT/bench_graph_leg.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06c17/bench_graph_leg.json') if l.startswith('{')][-1])
print('main %.0f q/s %.4f ms' % (d['value'], d['ms_per_step']))
print('rerank', d['rerank'])
print('graph', d['graph'])
print('ivf', {k: d['ivf'][k] for k in d['ivf'] if k in ('value', 'agreement_with_exhaustive_adc_top10', 'rerank')})
print('summary', json.dumps(d['summary']))
PY
