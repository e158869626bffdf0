#!/bin/bash
# round 5, evidence call A: PMC passes on the final kernel sources, rocprof kernel stats of the bench job, the bench lines.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp SGLANG_USE_AITER=0
export SGLANG_AMD_GIT_REV=${1:-unknown}
bash benchmarks/gpu_pmc_phases.sh > gpurun_out/pmc_phases.log 2>&1; tail -4 gpurun_out/pmc_phases.log | cut -c1-200
cp gpurun_out/r05_pmc.json profiles/r05_pmc.json 2>/dev/null && echo "pmc record in place"
bash benchmarks/gpu_prof.sh > gpurun_out/gpu_prof.log 2>&1; cp gpurun_out/kernel_stats.txt gpurun_out/r05_bench_kernel_stats.txt; head -14 gpurun_out/kernel_stats.txt | cut -c1-160
run() { name=$1; shift; timeout ${T:-480} python bench.py "$@" > gpurun_out/$name.json 2> gpurun_out/$name.log; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('gpurun_out/$name.json')); print(round(d['value'],1), 'tok/s', round(d['roofline']['frac'],4), 'step', round(d['roofline']['ms_per_decode_step'],3), 'ms', round(d['prefill_mfma']['frac'],3), 'prefill', 'traffic', d['roofline'].get('traffic'), 'ref_sched', d.get('reference_scheduler_tokens_per_s'))" 2>&1 | tail -1)"; }
run r05_bench_line_v2
run r05_bench_line_opsurface --operator-surface --no-cpu-baseline --no-parity
run r05_bench_line_fp8kv --kv-cache-dtype fp8_e4m3 --no-cpu-baseline
run r05_rank_8b_tp2 --rank-of 2
run r05_rank_8b_tp8 --rank-of 8
run r05_rank_70b_tp8 --model llama-3-70b --rank-of 8
