#!/bin/bash
# round 4: whole GPU suite + smoke on the current code
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q --durations=10 -x > gpurun_out/pytest_gpu.log 2>&1; tail -18 gpurun_out/pytest_gpu.log | cut -c1-220
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
