#!/bin/bash
# round-3 end-of-round check: the whole GPU suite (all failures listed, no -x), smoke, the default bench line
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/pytest_gpu.log 2>&1; tail -30 gpurun_out/pytest_gpu.log | cut -c1-220
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout 420 python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/bench.log; head -c 900 gpurun_out/r03_bench_line.json
