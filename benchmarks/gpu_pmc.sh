#!/bin/bash
# HBM traffic (PMC) of the dominant kernels, separate passes per counter (no trace domains mixed in).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -o run -- python $REPO/benchmarks/gemm_kernel_only.py > $REPO/gpurun_out/pmc_gemm_$C.log 2>&1)
  python benchmarks/summarize_pmc.py /tmp/pmc_$C gpurun_out/pmc_gemm_$C.txt wstream_gemm | tail -3
done
tail -1 gpurun_out/pmc_gemm_FETCH_SIZE.log
