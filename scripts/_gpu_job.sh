cd /root/repo
timeout 300 bash scripts/gpu_pmc_traffic.sh q8_1p25m --rows 1250000 --data lowrank --fused 2>&1 | tail -8
find gpurun_out -type f ! -name '*counter_collection.csv' -delete
for f in $(find gpurun_out -name '*counter_collection.csv'); do (head -1 $f; grep annlite $f) > $f.tmp; mv $f.tmp $f; done
