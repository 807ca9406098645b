cd /root/repo
timeout 900 bash scripts/gpu_profile.sh r02final --rows 10000000 --data lowrank --fused --iters 5 > /dev/null 2>&1
timeout 300 bash scripts/gpu_profile_bench.sh bench10m --steps 20 --warmup 5 --ivf-cells 0 > /dev/null 2>&1
# keep the small files only; counter files: the annlite kernels' rows
find gpurun_out -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' ! -name 'summary.txt' -delete
for f in $(find gpurun_out -name '*counter_collection.csv'); do (head -1 $f; grep annlite $f) > $f.tmp; mv $f.tmp $f; done
timeout 600 python bench.py > gpurun_out/bench_10m_n1.json 2> gpurun_out/bench_10m_n1.err
timeout 300 python bench.py --rows 1250000 --steps 40 --warmup 8 > gpurun_out/bench_1p25m_n1.json 2> gpurun_out/bench_1p25m_n1.err
timeout 300 python bench.py --rows 2000000 --dim 768 --m 64 --batch 256 --metric cosine --steps 10 --warmup 3 --ivf-cells 0 > gpurun_out/bench_config4_2m_n1.json 2> gpurun_out/bench_config4.err
du -sh gpurun_out
