#!/bin/bash
# round 5, evidence call C (re-run after the Python-side changes as v5): the whole GPU suite + smoke + the default bench line on the end-of-round tree
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp SGLANG_USE_AITER=0
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/pytest_gpu.log 2>&1; tail -14 gpurun_out/pytest_gpu.log | cut -c1-220
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 500 python bench.py > gpurun_out/r05_bench_line_v5.json 2> gpurun_out/r05_bench_line_v5.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_line_v5.json"))
rs = d.get("reference_scheduler") or {}
print("bench v5", round(d["value"]), d["unit"], "step", round(d["roofline"]["ms_per_decode_step"], 3), "ms =", round(d["roofline"]["frac"], 4), "of HBM; prefill", round(d["prefill_mfma"]["frac"], 3), "traffic", d["roofline"].get("traffic"), "pmc", d["pmc_record"], "| reference scheduler", rs.get("tokens_per_s"), rs.get("seconds_per_job"), rs.get("decode_step_ms_p50"), rs.get("triton_launches"))
PY
