#!/usr/bin/env bash
# Round 5, GPU call 15: HBM traffic (PMC: FETCH_SIZE / WRITE_SIZE / L2 hits, two passes each) of the legs' scan launches that had no kept pass:
# c2 (1M rows), the 1.25M-row shard, m32 (10M rows, M = 32).
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out/r05c15
bash scripts/gpu_pmc_traffic.sh r05_c2 --rows 1000000 --data lowrank --fused --valid > gpurun_out/r05c15/traffic_c2_1m.txt 2>&1; cat gpurun_out/r05c15/traffic_c2_1m.txt
bash scripts/gpu_pmc_traffic.sh r05_shard --rows 1250000 --data lowrank --fused --valid > gpurun_out/r05c15/traffic_shard_1p25m.txt 2>&1; cat gpurun_out/r05c15/traffic_shard_1p25m.txt
bash scripts/gpu_pmc_traffic.sh r05_m32 --rows 10000000 --m 32 --dsub 4 --data lowrank --fused --valid > gpurun_out/r05c15/traffic_m32_10m.txt 2>&1; cat gpurun_out/r05c15/traffic_m32_10m.txt
find gpurun_out -name "*.db" -delete 2>/dev/null; find gpurun_out -name "*kernel_trace.csv" -delete 2>/dev/null; find gpurun_out -name "*agent_info.csv" -delete 2>/dev/null
for f in $(find gpurun_out/traffic_r05_* -name '*counter_collection.csv'); do (head -1 $f; grep adc_scan $f) > $f.tmp; mv $f.tmp $f; done
