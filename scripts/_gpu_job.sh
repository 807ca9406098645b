cd /root/repo
timeout 600 bash scripts/gpu_profile.sh m64 --rows 2000000 --m 64 --batch 256 --data lowrank --fused --iters 4 > /dev/null 2>&1
find gpurun_out -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' ! -name 'summary.txt' -delete
for f in $(find gpurun_out -name '*counter_collection.csv'); do (head -1 $f; grep annlite $f) > $f.tmp; mv $f.tmp $f; done
grep -A10 "adc_scan_qfilter64" gpurun_out/prof_m64/summary.txt | head -60
