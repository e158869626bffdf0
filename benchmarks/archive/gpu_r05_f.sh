#!/bin/bash
# round 5, call 6: chunk-kernel forms incl. the five-per-CU looping instance, the default bench line with the reference_scheduler leg,
# and the xGMI / TP-hook tests after the per-communicator release-fence switch.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp SGLANG_USE_AITER=0
timeout 300 python benchmarks/r05_exp1_cascade_table_width.py > gpurun_out/r05_exp1b.log 2>&1; grep -v "^$" gpurun_out/r05_exp1b.log | tail -24 | cut -c1-200
cp gpurun_out/r05_exp1_cascade_table_width.json gpurun_out/r05_exp1b_cascade_forms.json
timeout 500 python bench.py 2>gpurun_out/bench.err | tail -1 > gpurun_out/r05_bench_line_v1.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench_line_v1.json"))
print("bench", round(d["value"]), d["unit"], "step", round(d["roofline"]["ms_per_decode_step"], 3), "ms =", round(d["roofline"]["frac"], 3), "of HBM; prefill", round(d["prefill_mfma"]["frac"], 3), "phase", {k: round(v, 1) for k, v in d["phase_ms"].items()})
print("reference_scheduler", json.dumps(d.get("reference_scheduler"))[:900])
print("kernel_table", json.dumps(d.get("kernel_table"))[:1500])
PY
timeout 900 python -m pytest tests/test_xgmi_all_reduce_gpu.py tests/test_tp_hooks_gpu.py tests/test_tp_sim_gpu.py tests/test_cascade_gpu.py -q -x 2>&1 | tail -5 | cut -c1-300
