#!/bin/bash
# round 5, call 8: re-run of call 7's test part after the hook-fallback fix
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp SGLANG_USE_AITER=0
timeout 1200 python -m pytest tests/test_model_hook_gpu.py tests/test_reference_model_gpu.py -q 2>&1 | tail -25 | cut -c1-900
timeout 900 python -m pytest tests/test_parity_full_gpu.py tests/test_layer_parity_gpu.py tests/test_moe_gpu.py tests/test_engine_gpu.py -q -k "mixtral or moe or Mixtral" 2>&1 | tail -8 | cut -c1-600
