#!/bin/bash
# round 5, call 5: reference-scheduler job after the RMSNorm hint fix / lm_head hook / bounded cascade grid, its step timeline, and the
# reference-stack GPU tests on the new pool / allocator classes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp SGLANG_USE_AITER=0
bash benchmarks/gpu_r05_b.sh v2 nobench 2>&1 | head -8
bash benchmarks/gpu_r05_c.sh v2 nobench 2>&1 | head -30 | cut -c1-170
timeout 1200 python -m pytest tests/test_reference_model_gpu.py tests/test_reference_objects_gpu.py tests/test_model_hook_gpu.py tests/test_kernels_gpu.py -q -x 2>&1 | tail -15 | cut -c1-400
