#!/bin/bash
# round 5: the hybrid fused decode layer at 65..128 rows (library gate_up + fused everything else) -- parity, then the two lines it moves
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp SGLANG_USE_AITER=0
timeout 600 python -m pytest tests/test_model_hook_gpu.py tests/test_layer_parity_gpu.py -m gpu -q -k "128 or 80 or 96 or llama-3-8b-2l-64 or 256" 2>&1 | tail -6
run() { name=$1; shift; timeout 500 python bench.py "$@" --no-cpu-baseline --no-reference-scheduler --no-parity > gpurun_out/$name.json 2> gpurun_out/$name.log
  python - "$name" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/{sys.argv[1]}.json"))
print(sys.argv[1], round(d["value"]), d["unit"], "step", round(d["roofline"]["ms_per_decode_step"], 3), "ms =", round(d["roofline"]["frac"], 4), "prefill", round(d["prefill_mfma"]["frac"], 3))
PY
}
run r05_bench_line_b128_hybrid --groups 8
run r05_rank_8b_tp2_hybrid --rank-of 2
