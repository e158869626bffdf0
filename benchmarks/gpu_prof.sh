#!/bin/bash
# rocprofv3 kernel stats of one bench step -> gpurun_out/kernel_stats.txt
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o runc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $REPO/gpurun_out/rocprof_bench.log 2>&1
cd $REPO && python benchmarks/summarize_rocprof.py /tmp/prof gpurun_out/kernel_stats.txt 60 > /dev/null 2>&1
grep '^{' gpurun_out/rocprof_bench.log | tail -1 | cut -c1-400
head -40 gpurun_out/kernel_stats.txt | cut -c1-200
