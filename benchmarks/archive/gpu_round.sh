#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprof kernel stats.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
( time timeout 600 python bench.py --steps 3 --warmup 1 ) > gpurun_out/bench.log 2>&1
( time timeout 300 python benchmarks/micro.py --json gpurun_out/micro.json ) > gpurun_out/micro.log 2>&1
REPO=$(pwd)
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o runc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/rocprof_bench.log 2>&1
cd $REPO && python benchmarks/summarize_rocprof.py /tmp/prof gpurun_out/kernel_stats.txt 45 > /dev/null 2>&1
tail -3 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/bench.log
