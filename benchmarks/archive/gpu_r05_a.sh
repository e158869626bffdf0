#!/bin/bash
# First GPU call of the next round: the whole GPU suite + smoke on the inherited tree, the default bench line, and BASELINE's job shape
# under the reference's Scheduler (overlap loop) with the plug-in -- the three numbers everything else is compared with.
#   gpurun --timeout 1500 -- bash benchmarks/gpu_r05_a.sh
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/pytest_gpu.log 2>&1; tail -22 gpurun_out/pytest_gpu.log | cut -c1-220
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 300 python bench.py 2>gpurun_out/bench.err | tail -1 > gpurun_out/bench_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_line.json"))
print("bench", round(d["value"]), d["unit"], "step", round(d["roofline"]["ms_per_decode_step"], 3), "ms =", round(d["roofline"]["frac"], 3), "of HBM")
PY
bash benchmarks/r04_reference_scheduler_job.sh 4,16,896,128,128 --overlap 2>&1 | grep -A3 '"timed"' | head -5
