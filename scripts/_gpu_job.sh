cd /root/repo
for sr in 8192 16384 32768 65536; do
for rows in 1250000 10000000; do
  ANNLITE_SEED_ROWS=$sr timeout 120 python bench.py --rows $rows --ivf-cells 0 --cpu-queries 0 --no-rerank --recall-queries 0 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('seed $sr rows $rows ms_per_step %.4f kernel_ms %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
done; done
