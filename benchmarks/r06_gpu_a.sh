#!/bin/bash
# round 6, GPU call A: cascade write-through A/B; per-position timeline of the decode layer inside the bench's step
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp SGLANG_USE_AITER=0
REPO=$(pwd)
timeout 300 python benchmarks/r06_cascade_ab.py gpurun_out/r06_exp3_cascade_wt.json 2>&1 | grep -v amdgpu.ids | tail -12
rm -rf /tmp/prof_seq
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seq -o run -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity > $REPO/gpurun_out/r06_seq_bench.log 2>&1
cd $REPO && python benchmarks/r06_step_sequence.py /tmp/prof_seq gpurun_out/r06_step_sequence_a.txt | cut -c1-200
grep '^{' gpurun_out/r06_seq_bench.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['ms_per_decode_step'])"
