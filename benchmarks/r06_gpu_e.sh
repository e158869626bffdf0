timeout 900 python -m pytest tests/test_sampler_gpu.py -x -q 2>&1 | tail -8
timeout 300 python benchmarks/r06_sampler_time.py gpurun_out/r06_sampler_time_v3.json 2>&1 | grep -v "ranges_\|widen\|softmax_us"
