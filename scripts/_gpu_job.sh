mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r2b/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2b/bench_10m.json 2> gpurun_out/r2b/bench_10m.err; tail -c 3500 gpurun_out/r2b/bench_10m.json; tail -3 gpurun_out/r2b/bench_10m.err
