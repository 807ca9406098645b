cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
for it in 1 2; do
for rows in 1250000 10000000; do
  timeout 120 python bench.py --rows $rows --ivf-cells 0 --cpu-queries 0 --no-rerank --recall-queries 0 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('rows $rows ms_per_step %.4f kernel_ms %.4f frac %.3f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac']))"
done; done
ANNLITE_DEBUG_COUNTERS=1 timeout 120 python scripts/prof_scan.py --rows 1250000 --data lowrank --fused --iters 6 2>&1 | grep -i "byte-table"
