#!/bin/bash
# round 5, call 2: rocprofv3 kernel trace of the reference-scheduler job (overlap loop) and of the bench's harness job -> per-step
# timelines (benchmarks/step_timeline.py): where do the reference stack's 4.9 ms per decode step go on the GPU?
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp SGLANG_USE_AITER=0
REPO=$(pwd)
tag=${1:-before}
rm -rf /tmp/prof_sched /tmp/prof_bench
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sched -o run -- python $REPO/tests/golden/ref_model.py --run scheduler --dims llama3_8b --job 4,16,896,128,128 --overlap --json $REPO/gpurun_out/r05_sched_${tag}_rocprof.json > $REPO/gpurun_out/prof_sched.log 2>&1
cd $REPO && python benchmarks/step_timeline.py /tmp/prof_sched gpurun_out/r05_sched_step_timeline_${tag}.txt | cut -c1-200
python benchmarks/summarize_rocprof.py /tmp/prof_sched gpurun_out/r05_sched_kernel_stats_${tag}.txt 60 > /dev/null 2>&1
if [ "${2:-bench}" = "bench" ]; then
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o run -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $REPO/gpurun_out/prof_bench.log 2>&1
cd $REPO && python benchmarks/step_timeline.py /tmp/prof_bench gpurun_out/r05_bench_step_timeline_${tag}.txt | cut -c1-200
fi
