timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r06_pytest_gpu_tail_b.txt
tail -5 gpurun_out/r06_pytest_gpu_tail_b.txt
