#!/bin/bash
# round-2 bundle A: PMC passes per phase first (the bench line quotes them), then the whole GPU suite (incl. the
# full-configuration parity tests), smoke, the default bench line (parity + cpu_baseline legs), the operator-surface
# line, rocprof kernel stats of one bench step, the Mixtral line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
bash benchmarks/gpu_pmc_phases.sh | head -40
if [ -s gpurun_out/r02_pmc.json ]; then cp gpurun_out/r02_pmc.json profiles/r02_pmc.json; fi
( time timeout 1500 python -m pytest tests -m gpu -x -q -s -k "parity_full or not parity_full" ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -a "passed\|failed\|error" gpurun_out/pytest_gpu.log | tail -3
( time timeout 300 python __graft_entry__.py smoke ) > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
( time timeout 900 python bench.py ) > gpurun_out/bench.log 2>&1
grep -a '^{' gpurun_out/bench.log | tail -1 > gpurun_out/bench_line.json
cut -c1-400 gpurun_out/bench_line.json; tail -5 gpurun_out/bench.log | cut -c1-300
( timeout 600 python bench.py --operator-surface --no-cpu-baseline --no-parity --no-kernel-roofline ) > gpurun_out/bench_opsurface.log 2>&1
grep -a '^{' gpurun_out/bench_opsurface.log | tail -1 > gpurun_out/bench_line_opsurface.json; cut -c1-300 gpurun_out/bench_line_opsurface.json
rm -rf /tmp/prof_bench
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_bench -o run -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-kernel-roofline > $OLDPWD/gpurun_out/rocprof_bench.log 2>&1)
python benchmarks/summarize_rocprof.py /tmp/prof_bench gpurun_out/kernel_stats.txt 45 | head -30
( timeout 900 python bench.py --model mixtral-8x7b --no-cpu-baseline --no-parity ) > gpurun_out/bench_mixtral.log 2>&1
grep -a '^{' gpurun_out/bench_mixtral.log | tail -1 > gpurun_out/bench_line_mixtral.json; cut -c1-400 gpurun_out/bench_line_mixtral.json; tail -3 gpurun_out/bench_mixtral.log | cut -c1-300
