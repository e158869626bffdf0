#!/bin/bash
# round 5, evidence call D: configs[4] (LLaVA-1.6 image + text prefill) and configs[2]'s model whole on one GPU, end-of-round tree
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp SGLANG_USE_AITER=0
timeout 300 python benchmarks/llava_prefill.py > gpurun_out/llava.log 2>&1; tail -3 gpurun_out/llava.log | cut -c1-400
cp gpurun_out/llava_prefill.json gpurun_out/r05_llava_prefill.json 2>/dev/null
timeout 900 python bench.py --model llama-3-70b --steps 2 --warmup 1 --no-cpu-baseline --no-reference-scheduler > gpurun_out/r05_bench_line_70b_tp1.json 2> gpurun_out/bench_70b.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_line_70b_tp1.json"))
print("70b tp1", round(d["value"]), d["unit"], "step", round(d["roofline"]["ms_per_decode_step"], 3), "ms =", round(d["roofline"]["frac"], 4), "prefill", round(d["prefill_mfma"]["frac"], 3), "ttft", d.get("ttft_p50_ms"))
PY
