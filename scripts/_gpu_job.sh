cd /root/repo
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -2
timeout 300 python bench.py --rows 2000000 --m 8 --ks 512 --steps 20 --warmup 5 --ivf-cells 0 > gpurun_out/bench_code16_m8_ks512_2m_n1.json 2> gpurun_out/err1.txt
timeout 300 python bench.py --rows 2000000 --m 8 --ks 768 --steps 20 --warmup 5 --ivf-cells 0 > gpurun_out/bench_code16_m8_ks768_2m_n1.json 2> gpurun_out/err2.txt
ANNLITE_NO_FAST_CODE16=1 timeout 300 python bench.py --rows 2000000 --m 8 --ks 768 --steps 5 --warmup 2 --ivf-cells 0 --cpu-queries 0 --no-rerank --recall-queries 0 > gpurun_out/bench_code16_m8_ks768_2m_generic.json 2> gpurun_out/err3.txt
timeout 300 bash scripts/gpu_profile_bench.sh code16 --rows 2000000 --m 8 --ks 768 --steps 20 --warmup 5 --ivf-cells 0 > /dev/null 2>&1
find gpurun_out/prof_code16 -type f ! -name '*kernel_stats.csv' ! -name 'summary.txt' -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_code16*.json')):
    try:
        r=json.load(open(f)); print(f, round(r['value']), r['ms_per_step'], r['roofline']['kernel'], r['roofline']['kernel_ms'], r['roofline']['frac'], (r.get('cpu_baseline') or {}).get('value'), (r.get('cpu_baseline') or {}).get('gpu_matches_cpu_bit_exact'), r.get('recall_at_10'), (r.get('rerank') or {}).get('recall_at_10'), (r.get('rerank') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
grep "adc_scan\|seed\|lut_" gpurun_out/prof_code16/summary.txt | cut -c1-160
