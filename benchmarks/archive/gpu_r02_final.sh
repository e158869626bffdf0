#!/bin/bash
# end-of-round check: the whole GPU suite, smoke, one bench line
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 420 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout 300 python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench.log; head -c 700 gpurun_out/bench_line.json
