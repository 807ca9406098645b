#!/usr/bin/env bash
# round 4, call 26 (the round's last GPU seconds): the bench's m32 leg command at 2M rows -- does the line come out, labelled as the byte-table kernel
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out/r04c26
timeout 16 python bench.py --m 32 --rows 2000000 --legs none --gpus 1 --streams 2 --steps 20 --warmup 5 --cpu-queries 0 --recall-queries 32 > gpurun_out/r04c26/bench_m32_2m.json 2>gpurun_out/r04c26/err.txt; echo rc=$?
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r04c26/bench_m32_2m.json') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); r=d['roofline']
    print(d['config']['workload'], '| q/s %.0f ms/step %.4f | %s kernel_ms %.4f frac %.3f per_clk %s choice %s | recall %s' % (d['value'], d['ms_per_step'], r['kernel'], r['kernel_ms'], r['frac'], r['lookups_per_clk_per_cu'], r['kernel_choice'], d.get('recall_at_10')))
else:
    print(open('gpurun_out/r04c26/err.txt').read()[-600:])
PY
