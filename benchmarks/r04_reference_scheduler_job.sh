#!/bin/bash
# BASELINE.json configs[1]'s job shape (4 x 16 requests, 896 shared + 128 own tokens in, 128 out, greedy) under the REFERENCE'S
# SCHEDULER with the plug-in loaded (tests/golden/ref_model.py run_scheduler_job): leaders arrive first, the other 60 requests once the
# leaders' prefill is in the radix tree.  Llama-3-8B architecture, dummy weights; event_loop_normal's body (no overlap scheduling).
#   gpurun -- bash benchmarks/r04_reference_scheduler_job.sh
set -e
job=${1:-4,16,896,128,128}
out=gpurun_out/reference_scheduler_job_${job//,/_}.json
SGLANG_USE_AITER=0 timeout 400 python tests/golden/ref_model.py --run scheduler --dims llama3_8b --job "$job" --json "$out" 2>gpurun_out/reference_scheduler_job.err | tail -60
