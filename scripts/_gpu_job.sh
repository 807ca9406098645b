cd /root/repo
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -2
timeout 300 python bench.py --rows 2000000 --dim 768 --m 64 --batch 256 --metric cosine --steps 10 --warmup 3 --ivf-cells 0 > gpurun_out/bench_config4_2m_n1.json 2> gpurun_out/bench_config4.err
python -c "import json; r=json.load(open('gpurun_out/bench_config4_2m_n1.json')); print('config4', r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac'], r['rerank']['value'], r['rerank']['recall_at_10'])"
timeout 120 python bench.py --rows 10000000 --ivf-cells 0 --cpu-queries 0 --no-rerank --recall-queries 0 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('10M ms_per_step %.4f kernel_ms %.4f frac %.3f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac']))"
