#!/bin/bash
# usage: scripts/gpu_ivf_ablate.sh "<cells> <probes>" ...   (on the GPU box) -- pruned-search time per stage; ablations
# through ANNLITE_DEBUG_SKIP: 4 = no candidate handling at all, 1 = no exact gathers, 2 = no list insertion
for c in "$@"; do set -- $c; for e in ${ABLATE:-X=1 ANNLITE_DEBUG_SKIP=4}; do echo "== cells $1 probes $2 $e"
env $e timeout 250 python scripts/bench_ivf.py --cells $1 --probes $2 --no-rerank --reps 5 2>&1 | grep n_probe | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['n_probe'], r['tiles_used'], round(r['ms'],3), round(r['recall_vs_exhaustive_adc'],4), json.dumps({k:(round(v,4) if isinstance(v,float) else v) for k,v in r['stages_ms'].items()}))"
done; done
