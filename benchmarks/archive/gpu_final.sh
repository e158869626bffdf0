#!/bin/bash
# End-of-milestone check on one GPU box: full parity suite, smoke(), default bench line, rocprof stats, PMC passes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
( time timeout 300 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1
( time timeout 600 python bench.py ) > gpurun_out/bench.log 2>&1
grep '^{' gpurun_out/bench.log | tail -1 > gpurun_out/bench_line.json
TOPN=3 bash benchmarks/gpu_prof.sh > /dev/null 2>&1
bash benchmarks/gpu_pmc.sh > /dev/null 2>&1
tail -4 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; cut -c1-300 gpurun_out/bench_line.json
