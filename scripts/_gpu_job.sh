# scratch: the command of the last gpurun call of a work session (see scripts/gpu_profile*.sh for the kept recipes)
cd /root/repo
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -2
