mkdir -p gpurun_out/r2e
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r2e/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2e/bench_10m.json 2> gpurun_out/r2e/bench_10m.err; python -c "
import json; r=json.load(open('gpurun_out/r2e/bench_10m.json')); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac'], r['cpu_baseline']['gpu_matches_cpu_bit_exact'], r['rerank']['value'], r['rerank']['recall_at_10'])"
timeout 300 python bench.py --rows 1250000 --steps 40 --warmup 5 --ivf-cells 0 --cpu-queries 0 --no-rerank > gpurun_out/r2e/bench_1p25m.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/r2e/bench_1p25m.json')); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'])"
timeout 300 bash scripts/gpu_profile_bench.sh r02e_bench10m --steps 20 --warmup 5 --ivf-cells 0 2>&1 | head -8 | cut -c1-200
