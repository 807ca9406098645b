cd /root/repo
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 120 python bench.py --rows 10000000 --ivf-cells 0 --cpu-queries 0 --no-rerank --recall-queries 0 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('10M ms_per_step %.4f kernel_ms %.4f frac %.3f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac']))"
