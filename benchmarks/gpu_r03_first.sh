#!/bin/bash
# First GPU call of the next round (about one GPU-minute): the kernel variant prepared at the end of round 2 without
# hardware time left to run it -- extend attention with its Q^T fragments in registers (SGL_AMD_EXTEND_QREG=1) --
# correctness against the oracle and the default kernel's bits, then the A/B timing on the bench's prefill shapes.
set -u
mkdir -p gpurun_out
SGLANG_AMD_RUN_EXPERIMENTS=1 timeout 120 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k q_in_registers 2>&1 | tail -3
echo "== default";  timeout 60 python benchmarks/r02_exp10_ext_time.py
echo "== Q^T in registers"; SGL_AMD_EXTEND_QREG=1 timeout 60 python benchmarks/r02_exp10_ext_time.py
