#!/bin/bash
# round 5, call 4: the chunk kernel vs the request table's width (single-shot vs looping forms), the cascade parity tests, and the
# reference-scheduler job with the native bookkeeping hooks + the bounded grid.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp SGLANG_USE_AITER=0
timeout 400 python benchmarks/r05_exp1_cascade_table_width.py > gpurun_out/r05_exp1.log 2>&1; tail -24 gpurun_out/r05_exp1.log | cut -c1-200
timeout 600 python -m pytest tests/test_cascade_gpu.py tests/test_engine_gpu.py -q -x 2>&1 | tail -3
bash benchmarks/gpu_r05_b.sh hooks nobench
bash benchmarks/gpu_r05_c.sh hooks2 nobench
