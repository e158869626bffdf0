#!/bin/bash
# round 5, call 7: the sparse-MoE form of the fused decode hook (tests + Mixtral bench lines), the reference-stack tests with the
# scheduler-level logit-band / zero-Triton assertions.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp SGLANG_USE_AITER=0
timeout 900 python -m pytest tests/test_model_hook_gpu.py tests/test_reference_model_gpu.py -q -x 2>&1 | tail -12 | cut -c1-600
timeout 900 python -m pytest tests/test_parity_full_gpu.py tests/test_layer_parity_gpu.py tests/test_moe_gpu.py -q -x -k "mixtral or moe or Mixtral" 2>&1 | tail -6 | cut -c1-400
run() { name=$1; shift; timeout ${T:-420} python bench.py "$@" > gpurun_out/$name.json 2> gpurun_out/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('gpurun_out/$name.json')); print(round(d['value'],1), 'tok/s', round(d['roofline']['frac'],4), 'step', round(d['roofline']['ms_per_decode_step'],3), 'ms', round(d['prefill_mfma']['frac'],3), 'prefill')" 2>&1 | tail -1)"; }
run r05_bench_line_mixtral_tp1 --model mixtral-8x7b --no-cpu-baseline --no-reference-scheduler
run r05_rank_mixtral_tp2 --model mixtral-8x7b --rank-of 2
