cd /root/repo
timeout 300 bash scripts/gpu_profile_bench.sh bench10m --ivf-cells 0 > /dev/null 2>&1
find gpurun_out -type f ! -name '*kernel_stats.csv' ! -name 'summary.txt' -delete
timeout 600 python bench.py > gpurun_out/bench_10m_n1.json 2> gpurun_out/bench_10m_n1.err
timeout 300 python bench.py --rows 1250000 > gpurun_out/bench_1p25m_n1.json 2> gpurun_out/bench_1p25m_n1.err
timeout 300 python bench.py --rows 2000000 --dim 768 --m 64 --batch 256 --metric cosine --steps 10 --warmup 3 --ivf-cells 0 > gpurun_out/bench_config4_2m_n1.json 2> gpurun_out/bench_config4.err
