#!/bin/bash
# the reference's bench_one_batch with the plug-in at small batches (Llama-3-8B dimensions, dummy weights)
mkdir -p gpurun_out
for shape in "1,1024,128" "16,1024,128"; do
  out=gpurun_out/r05_one_batch_${shape//,/_}.json
  SGLANG_USE_AITER=0 timeout 300 python tests/golden/ref_model.py --run latency --dims llama3_8b --shape "$shape" --json "$out" \
    2>gpurun_out/r05_one_batch_small.err | grep -E "prefill_latency|median_decode|graph_replays" | tr '\n' ' '
  echo " <- shape $shape"
done
