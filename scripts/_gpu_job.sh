cd /root/repo
for rk in 16 32; do
timeout 300 python bench.py --rows 10000000 --steps 12 --warmup 3 --ivf-cells 0 --cpu-queries 0 --rerank-k $rk 2>gpurun_out/err_rk$rk.txt | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('rerank_k $rk:', r['rerank'])" || tail -3 gpurun_out/err_rk$rk.txt
done
