#!/bin/bash
# round 5, evidence call B: the whole GPU suite + smoke on the end-of-round tree, clean rocprof kernel stats of the bench job, the
# default bench line (reference_scheduler leg with the committed GEMM selections loaded by the platform).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp SGLANG_USE_AITER=0
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/pytest_gpu.log 2>&1; tail -20 gpurun_out/pytest_gpu.log | cut -c1-220
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
bash benchmarks/gpu_prof.sh --no-parity --no-reference-scheduler > gpurun_out/gpu_prof.log 2>&1; cp gpurun_out/kernel_stats.txt gpurun_out/r05_bench_kernel_stats.txt; head -12 gpurun_out/kernel_stats.txt | cut -c1-160
timeout 500 python bench.py > gpurun_out/r05_bench_line_v3.json 2> gpurun_out/r05_bench_line_v3.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_line_v3.json"))
rs = d.get("reference_scheduler") or {}
print("bench v3", round(d["value"]), d["unit"], "step", round(d["roofline"]["ms_per_decode_step"], 3), "ms =", round(d["roofline"]["frac"], 4), "of HBM; prefill", round(d["prefill_mfma"]["frac"], 3), "traffic", d["roofline"].get("traffic"), "| reference scheduler", rs.get("tokens_per_s"), rs.get("seconds_per_job"), rs.get("decode_step_ms_p50"), rs.get("triton_launches"))
PY
