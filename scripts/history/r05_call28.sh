#!/usr/bin/env bash
# Round 5, GPU call 28: HBM traffic of the k = 50 launch after the seed rows were scaled with k
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; mkdir -p gpurun_out/r05c28
bash scripts/gpu_pmc_traffic.sh r05_k50_final --rows 10000000 --data lowrank --fused --valid --k 50 > gpurun_out/r05c28/traffic_k50_10m.txt 2>&1; cat gpurun_out/r05c28/traffic_k50_10m.txt
find gpurun_out -name "*.db" -delete 2>/dev/null; find gpurun_out -name "*kernel_trace.csv" -delete 2>/dev/null; find gpurun_out -name "*agent_info.csv" -delete 2>/dev/null
for f in $(find gpurun_out/traffic_r05_* -name '*counter_collection.csv'); do (head -1 $f; grep adc_scan $f) > $f.tmp; mv $f.tmp $f; done
