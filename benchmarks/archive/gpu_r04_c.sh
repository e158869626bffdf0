#!/bin/bash
# round 4, evidence call 1: PMC passes over the bench job + rocprof kernel stats of one bench step (end-of-round sources)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
export SGLANG_AMD_GIT_REV=${SGLANG_AMD_GIT_REV:-unknown}
bash benchmarks/gpu_pmc_phases.sh > gpurun_out/pmc.log 2>&1; head -8 gpurun_out/pmc.log | cut -c1-200
bash benchmarks/gpu_prof.sh --no-parity > gpurun_out/prof.log 2>&1
cp gpurun_out/kernel_stats.txt gpurun_out/r04_bench_kernel_stats.txt; head -16 gpurun_out/r04_bench_kernel_stats.txt | cut -c1-170
