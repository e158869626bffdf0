#!/bin/bash
# round 6, GPU call B: prefill region of the bench job (busy vs wall, gaps) and the decode layer's per-position timeline
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp SGLANG_USE_AITER=0
REPO=$(pwd)
rm -rf /tmp/prof_seq
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seq -o run -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-kernel-roofline --no-reference-scheduler > $REPO/gpurun_out/r06_seq_bench.log 2>&1
cd $REPO && python benchmarks/r06_step_sequence.py /tmp/prof_seq gpurun_out/r06_prefill_region.txt x prefill | cut -c1-200
python benchmarks/r06_step_sequence.py /tmp/prof_seq gpurun_out/r06_step_sequence_b.txt | cut -c1-160 | tail -14
