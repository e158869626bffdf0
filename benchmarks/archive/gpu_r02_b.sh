#!/bin/bash
# quick loop: the GEMM / engine tests, then rocprof kernel stats of one bench step
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_wstream_gemm_gpu.py tests/test_engine_gpu.py tests/test_tp_sim_gpu.py -x -q 2>&1 | tail -4
rm -rf /tmp/prof_bench
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_bench -o run -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-kernel-roofline > $OLDPWD/gpurun_out/rocprof_bench.log 2>&1)
python benchmarks/summarize_rocprof.py /tmp/prof_bench gpurun_out/kernel_stats.txt 45 | head -16
