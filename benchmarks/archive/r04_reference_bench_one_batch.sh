#!/bin/bash
# The reference's own latency benchmark (python -m sglang.bench_one_batch --load-format dummy: benchmark/one_batch.py latency_test)
# on the reference's ModelRunner with the plug-in loaded, Llama-3-8B architecture, from the staged copy of the reference
# (tests/golden/ref_model.py run_latency).  Static batch, distinct prompts, decode steps = replays of the reference's decode graphs.
#   gpurun -- bash benchmarks/r04_reference_bench_one_batch.sh "64,1024,128"
set -e
shape=${1:-64,1024,128}
out=gpurun_out/reference_bench_one_batch_${shape//,/_}.json
SGLANG_USE_AITER=0 timeout 400 python tests/golden/ref_model.py --run latency --dims llama3_8b --shape "$shape" --json "$out" 2>gpurun_out/reference_bench_one_batch.err | tail -40
