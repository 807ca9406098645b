mkdir -p gpurun_out/r2d
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r2d/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2d/bench_10m.json 2> gpurun_out/r2d/bench_10m.err; python -c "
import json; r=json.load(open('gpurun_out/r2d/bench_10m.json')); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac'], r['cpu_baseline']['gpu_matches_cpu_bit_exact'], r['rerank'])"
timeout 300 python bench.py --rows 1250000 --steps 40 --warmup 5 --ivf-cells 0 --cpu-queries 0 --no-rerank > gpurun_out/r2d/bench_1p25m.json 2>/dev/null; python -c "
import json; r=json.load(open('gpurun_out/r2d/bench_1p25m.json')); print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'])"
timeout 300 python bench.py --dim 768 --m 64 --batch 256 --metric cosine --rows 2000000 --steps 10 --warmup 3 --ivf-cells 0 --cpu-queries 32 --cpu-repeats 1 > gpurun_out/r2d/bench_c4_2m.json 2>gpurun_out/r2d/bench_c4_2m.err; python -c "
import json; r=json.load(open('gpurun_out/r2d/bench_c4_2m.json')); print('C4 2M', r['value'], r['ms_per_step'], r['roofline']['frac'], r['cpu_baseline']['gpu_matches_cpu_bit_exact'], r['recall_at_10'])"; tail -2 gpurun_out/r2d/bench_c4_2m.err
