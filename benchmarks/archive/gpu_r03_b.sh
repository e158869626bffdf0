#!/bin/bash
# round-3 evidence on the end-of-round code: whole GPU suite, smoke, bench lines, kernel stats, PMC passes
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/pytest_gpu.log 2>&1; tail -14 gpurun_out/pytest_gpu.log | cut -c1-200
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 420 python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/bench.log; head -c 600 gpurun_out/r03_bench_line.json; echo
timeout 300 python bench.py --operator-surface --no-cpu-baseline --no-parity > gpurun_out/r03_bench_line_opsurface.json 2> gpurun_out/bench_os.log; head -c 300 gpurun_out/r03_bench_line_opsurface.json; echo
bash benchmarks/gpu_prof.sh --no-parity > gpurun_out/prof.log 2>&1; head -12 gpurun_out/kernel_stats.txt | cut -c1-150
bash benchmarks/gpu_pmc_phases.sh > gpurun_out/pmc.log 2>&1; head -8 gpurun_out/pmc.log | cut -c1-200
