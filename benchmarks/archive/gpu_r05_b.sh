#!/bin/bash
# round 5, call 1: the inherited tree's reference-scheduler job (overlap loop) -- plain, then under cProfile (host-time split,
# profiles/r05_host_step*.txt) -- and the default bench line.
#   gpurun --timeout 900 -- bash benchmarks/gpu_r05_b.sh
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp SGLANG_USE_AITER=0
tag=${1:-before}
job=4,16,896,128,128
timeout 400 python tests/golden/ref_model.py --run scheduler --dims llama3_8b --job $job --overlap --json gpurun_out/r05_sched_${tag}.json > /dev/null 2> gpurun_out/r05_sched_${tag}.err
python - <<PY
import json
d = json.load(open("gpurun_out/r05_sched_${tag}.json"))
print("sched ${tag}", round(d["timed"]["output_tokens_per_s"]), "tok/s", round(d["timed"]["seconds"], 4), "s", d["timed"]["batches_run"], "triton", d.get("triton_launches_in_the_timed_job"), d.get("triton_kernels_in_the_timed_job"), d.get("kv_pool_class"), d.get("allocator_class"), d.get("graph_runner"))
PY
REF_SCHED_CPROFILE=gpurun_out/r05_host_step_${tag}.txt timeout 400 python tests/golden/ref_model.py --run scheduler --dims llama3_8b --job $job --overlap --json gpurun_out/r05_sched_${tag}_profiled.json > /dev/null 2> gpurun_out/r05_sched_${tag}_profiled.err
head -12 gpurun_out/r05_host_step_${tag}.txt
if [ "${2:-bench}" = "bench" ]; then
timeout 300 python bench.py 2>gpurun_out/bench.err | tail -1 > gpurun_out/r05_bench_line_${tag}.json
python - <<PY
import json
d = json.load(open("gpurun_out/r05_bench_line_${tag}.json"))
print("bench", round(d["value"]), d["unit"], "step", round(d["roofline"]["ms_per_decode_step"], 3), "ms =", round(d["roofline"]["frac"], 3), "of HBM")
PY
fi
