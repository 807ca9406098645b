cd /root/repo
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench_10m_n1.json 2> gpurun_out/bench_10m_n1.err
python -c "
import json; r=json.load(open('gpurun_out/bench_10m_n1.json')); print(round(r['value']), r['ms_per_step'], r['roofline']['kernel'], r['roofline']['kernel_ms'], r['roofline']['frac'], r['roofline']['hbm'], r['cpu_baseline']['value'], r['cpu_baseline']['gpu_matches_cpu_bit_exact'], r['recall_at_10'], r['rerank']['value'], r['rerank']['recall_at_10'], r['ivf']['value'])"
