cd /root/repo
for st in 1 2; do
for rows in 1250000 10000000; do
  timeout 120 python bench.py --rows $rows --streams $st --ivf-cells 0 --cpu-queries 0 --no-rerank --recall-queries 0 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('streams $st rows $rows ms_per_step %.4f kernel_ms %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
done; done
