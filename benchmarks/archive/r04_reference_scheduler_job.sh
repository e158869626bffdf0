#!/bin/bash
# BASELINE.json configs[1]'s job shape (4 x 16 requests, 896 shared + 128 own tokens in, 128 out, greedy) under the REFERENCE'S
# SCHEDULER with the plug-in loaded (tests/golden/ref_model.py run_scheduler_job): leaders arrive first, the other 60 requests once the
# leaders are decoding (their prompts are in the radix tree).  Llama-3-8B architecture, dummy weights; the scheduler's own run_event_loop().
#   gpurun -- bash benchmarks/r04_reference_scheduler_job.sh [job] [--overlap]
set -e
job=${1:-4,16,896,128,128}
loop=${2:-}           # "--overlap": event_loop_overlap (the server default) instead of event_loop_normal
out=gpurun_out/reference_scheduler_job_${job//,/_}${loop:+_overlap}.json
SGLANG_USE_AITER=0 timeout 400 python tests/golden/ref_model.py --run scheduler --dims llama3_8b --job "$job" $loop --json "$out" 2>gpurun_out/reference_scheduler_job.err | tail -60
