timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-queries 0 --ivf-cells 256 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], 'rerank', r['rerank']['value'], r['rerank']['recall_at_10'], 'ivf', r['ivf']['value'], r['ivf']['rerank']['value'])"
timeout 300 python bench.py --dim 768 --m 64 --batch 256 --metric cosine --rows 2000000 --steps 10 --warmup 3 --ivf-cells 0 --cpu-queries 0 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('C4 2M', r['value'], r['ms_per_step'], r['roofline']['frac'])"
