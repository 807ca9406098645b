cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
for sr in 16384 32768 65536 131072; do
for rows in 1250000 10000000; do
  ANNLITE_SEED_ROWS=$sr timeout 120 python bench.py --rows $rows --steps 40 --warmup 8 --ivf-cells 0 --cpu-queries 0 --no-rerank --recall-queries 0 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('seed $sr rows $rows ms_per_step %.4f kernel_ms %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
done; done
