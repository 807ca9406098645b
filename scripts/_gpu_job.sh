mkdir -p gpurun_out/r2c
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r2c/pytest_gpu.log
