mkdir -p gpurun_out/r2b
timeout 300 bash scripts/gpu_profile_bench.sh r02_bench10m --steps 20 --warmup 5 --ivf-cells 0 2>&1 | tail -12
timeout 400 bash scripts/gpu_profile.sh r02_scan10m --data lowrank --rows 10000000 --valid --fused --iters 5 2>&1 | tail -45
timeout 300 python bench.py --rows 1250000 --steps 40 --warmup 5 --ivf-cells 0 --cpu-queries 0 --no-rerank > gpurun_out/r2b/bench_1p25m.json 2>/dev/null; tail -c 1500 gpurun_out/r2b/bench_1p25m.json
