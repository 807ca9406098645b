cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "uint16 or wide" 2>&1 | tail -3
timeout 300 python bench.py --rows 2000000 --m 8 --ks 512 --steps 20 --warmup 5 --ivf-cells 0 > gpurun_out/bench_code16_m8_ks512_2m_n1.json 2> gpurun_out/err1.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_code16*.json')):
    try:
        r=json.load(open(f)); print(f, round(r['value']), r['ms_per_step'], r['roofline']['kernel'], r['roofline']['kernel_ms'], r['roofline']['frac'], (r.get('cpu_baseline') or {}).get('value'), (r.get('cpu_baseline') or {}).get('gpu_matches_cpu_bit_exact'), r.get('recall_at_10'), (r.get('rerank') or {}).get('recall_at_10'), (r.get('rerank') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
