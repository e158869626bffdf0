timeout 900 python -m pytest tests/test_sampler_gpu.py -x -q 2>&1 | tail -5
python benchmarks/r06_sampler_prof.py
TOPN=4 bash benchmarks/prof_cmd.sh r06_sampler python $(pwd)/benchmarks/r06_sampler_prof.py
timeout 300 python benchmarks/r06_sampler_time.py gpurun_out/r06_sampler_time_v2.json 2>&1 | grep -v "ranges_\|widen\|softmax_us"
