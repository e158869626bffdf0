#!/bin/bash
# round 5: configurations of the reference's scheduler the GPU suite does not cover yet, run for defects (tokens vs the oracle, logit band)
export TMPDIR=/tmp SGLANG_USE_AITER=0
mkdir -p gpurun_out
run() { tag=$1; shift; timeout 300 python tests/golden/ref_model.py --run scheduler --overlap --json gpurun_out/var_$tag.json "$@" > gpurun_out/var_$tag.out 2> gpurun_out/var_$tag.err; rc=$?; python - <<PY 2>&1 | tail -1
import json
try:
    d = json.load(open("gpurun_out/var_$tag.json"))
    lb = d.get("logit_band") or {}
    print("$tag rc=$rc", d["oracle"], "band", {k: round(v, 5) if isinstance(v, float) else v for k, v in lb.items() if k in ("rows_compared", "product_rms_err", "reference_rms_err", "product_max_err", "reference_max_err")},
          d["timed"]["batches_run"], "replays", d["graph_replays_in_the_timed_job"], "triton", d["triton_launches_in_the_timed_job"], d.get("spec") and d["spec"]["forward_modes_in_the_timed_job"])
except Exception as e:
    print("$tag rc=$rc FAILED", type(e).__name__, e)
PY
[ $rc -ne 0 ] && tail -6 gpurun_out/var_$tag.err | cut -c1-250; }
run eager --server-args '{"disable_cuda_graph": true}'
run small_graph --job 2,3,16,8,6 --server-args '{"cuda_graph_max_bs_decode": 2}'
run fp8kv --server-args '{"kv_cache_dtype": "fp8_e4m3"}'
run qwen2 --dims tiny_qwen2
run no_radix --server-args '{"disable_radix_cache": true}'
run paged_spec --spec-ngram 4 --job 2,2,32,16,12 --server-args '{"page_size": 16}'
run spec_mixtral --spec-ngram 3 --dims tiny_mixtral --job 2,2,16,8,10
run tp2 --tp 2
