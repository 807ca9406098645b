#!/usr/bin/env bash
# Round 5, GPU call 39: HBM traffic (PMC passes, kernel-trace only) of the headline scan, the shard and the 1M-row table on the final tree
# (seed rows spread, first epoch end at step 255, table target 88): what profiles/traffic.json holds for adc_scan_q8_kernel at M = 16
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; mkdir -p gpurun_out/r05c39
bash scripts/gpu_pmc_traffic.sh r05_final_10m --rows 10000000 --data lowrank --fused --valid > gpurun_out/r05c39/traffic_final_10m.txt 2>&1; cat gpurun_out/r05c39/traffic_final_10m.txt
bash scripts/gpu_pmc_traffic.sh r05_final_shard --rows 1250000 --data lowrank --fused --valid > gpurun_out/r05c39/traffic_final_shard_1p25m.txt 2>&1; cat gpurun_out/r05c39/traffic_final_shard_1p25m.txt
bash scripts/gpu_pmc_traffic.sh r05_final_c2 --rows 1000000 --data lowrank --fused --valid > gpurun_out/r05c39/traffic_final_c2_1m.txt 2>&1; cat gpurun_out/r05c39/traffic_final_c2_1m.txt
find gpurun_out -name "*.db" -delete 2>/dev/null; find gpurun_out -name "*kernel_trace.csv" -delete 2>/dev/null; find gpurun_out -name "*agent_info.csv" -delete 2>/dev/null
for f in $(find gpurun_out/traffic_r05_* -name '*counter_collection.csv'); do (head -1 $f; grep "adc_scan_q8" $f) > $f.tmp; mv $f.tmp $f; done
du -sh gpurun_out
