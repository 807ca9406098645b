#!/usr/bin/env bash
# Round 5, GPU call 20: HBM traffic of the M = 32 launch under the new plan (8 slices, slice-per-XCD)
set -u
cd "$(dirname "$0")/../.."; rm -rf gpurun_out/*; mkdir -p gpurun_out/r05c20
bash scripts/gpu_pmc_traffic.sh r05_m32_s8 --rows 10000000 --m 32 --dsub 4 --data lowrank --fused --valid > gpurun_out/r05c20/traffic_m32_10m_8slices.txt 2>&1; cat gpurun_out/r05c20/traffic_m32_10m_8slices.txt
find gpurun_out -name "*.db" -delete 2>/dev/null; find gpurun_out -name "*kernel_trace.csv" -delete 2>/dev/null; find gpurun_out -name "*agent_info.csv" -delete 2>/dev/null
for f in $(find gpurun_out/traffic_r05_* -name '*counter_collection.csv'); do (head -1 $f; grep adc_scan $f) > $f.tmp; mv $f.tmp $f; done
