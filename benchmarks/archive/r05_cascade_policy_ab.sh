#!/bin/bash
# Shared-prefix decode plan against the plain paged decode kernel on batches WITHOUT a shared prefix: the reference's own latency
# benchmark (bench_one_batch: distinct prompts, decode steps = replays of the reference's decode graphs) with SGLANG_AMD_CASCADE=0 / 1
# on one box.   gpurun -- bash benchmarks/r05_cascade_policy_ab.sh
mkdir -p gpurun_out
for shape in "64,1024,128" "16,4096,64"; do
  for c in 0 1; do
    out=gpurun_out/r05_one_batch_${shape//,/_}_cascade$c.json
    SGLANG_AMD_CASCADE=$c SGLANG_USE_AITER=0 timeout 400 python tests/golden/ref_model.py --run latency --dims llama3_8b --shape "$shape" --json "$out" \
      2>gpurun_out/r05_one_batch_cascade$c.err | grep -E "prefill_latency|median_decode|graph_replays" | tr '\n' ' '
    echo " <- shape $shape cascade $c"
  done
done
timeout 600 python -m pytest tests/test_reference_model_gpu.py -q -m gpu -k "no-radix or eager or small-graph" 2>&1 | tail -3
