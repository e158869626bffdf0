#!/bin/bash
# round 4, evidence call 2: bench lines on the end-of-round code with the PMC record current
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; timeout ${T:-420} python bench.py "$@" > gpurun_out/$name.json 2> gpurun_out/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('gpurun_out/$name.json')); print(round(d['value'],1), 'tok/s', round(d['roofline']['frac'],4), 'step', round(d['prefill_mfma']['frac'],3), 'prefill')" 2>&1 | tail -1)"; }
run r04_bench_line_v2
run r04_bench_line_opsurface --operator-surface --no-cpu-baseline --no-parity
run r04_bench_line_mixtral_tp1 --model mixtral-8x7b --no-cpu-baseline
run r04_bench_line_fp8kv --kv-cache-dtype fp8_e4m3 --no-cpu-baseline
run r04_rank_8b_tp2 --rank-of 2
run r04_rank_8b_tp4 --rank-of 4
run r04_rank_8b_tp8 --rank-of 8
run r04_rank_70b_tp8 --model llama-3-70b --rank-of 8
run r04_rank_mixtral_tp2 --model mixtral-8x7b --rank-of 2
timeout 300 python benchmarks/moe_prefill_micro.py > gpurun_out/moe_micro.log 2>&1; tail -4 gpurun_out/moe_micro.log | cut -c1-200
timeout 300 python benchmarks/llava_prefill.py > gpurun_out/llava.log 2>&1; tail -3 gpurun_out/llava.log | cut -c1-300
