#!/bin/bash
# BASELINE.json configs[1]'s job shape (4 x 16 requests, 896 shared + 128 own tokens in, 128 out, greedy) through the REFERENCE'S
# ModelRunner / ScheduleBatch / DecodeCudaGraphRunner / sampler with the plug-in loaded (tests/golden/ref_model.py
# run_shared_prefix_job; Llama-3-8B architecture, dummy weights).  Beside bench.py's number for the same job on this package's harness.
#   gpurun -- bash benchmarks/r04_reference_runner_shared_prefix.sh [job] [--radix]
set -e
job=${1:-4,16,896,128,128}
radix=${2:-}          # "--radix": the prefixes come out of the reference's real RadixCache (match_prefix / cache_unfinished_req)
out=gpurun_out/reference_runner_shared_prefix_${job//,/_}${radix:+_radix}.json
SGLANG_USE_AITER=0 timeout 400 python tests/golden/ref_model.py --run shared-prefix --dims llama3_8b --job "$job" $radix --json "$out" 2>gpurun_out/reference_runner_shared_prefix.err | grep -v '"what"\|"identical"\|"ref_rms"' | tail -45
