#!/bin/bash
export TMPDIR=/tmp SGLANG_USE_AITER=0
mkdir -p gpurun_out
run() { tag=$1; shift; timeout 300 python tests/golden/ref_model.py --run scheduler --overlap --json gpurun_out/var_$tag.json "$@" > gpurun_out/var_$tag.out 2> gpurun_out/var_$tag.err; rc=$?; python - <<PY 2>&1 | tail -1
import json
try:
    d = json.load(open("gpurun_out/var_$tag.json"))
    lb = d.get("logit_band") or {}
    print("$tag rc=$rc", d["oracle"], "band", {k: round(v, 5) if isinstance(v, float) else v for k, v in lb.items() if k in ("rows_compared", "rows_expected", "product_rms_err", "reference_rms_err", "product_max_err", "reference_max_err", "clear_rows", "argmax_agree_on_clear_rows")},
          d["timed"]["batches_run"], "replays", d["graph_replays_in_the_timed_job"], "triton", d["triton_launches_in_the_timed_job"], d.get("sampling"), d["timed"]["cached_tokens_of_others"])
except Exception as e:
    print("$tag rc=$rc FAILED", type(e).__name__, e)
PY
[ $rc -ne 0 ] && tail -6 gpurun_out/var_$tag.err | cut -c1-250; }
run long_shared --job 2,4,160,40,48
run long_shared_paged --job 2,4,160,40,48 --server-args '{"page_size": 16}'
run long_shared_chunked --job 2,4,300,40,20 --server-args '{"chunked_prefill_size": 128}'
run sampling --job 2,2,16,8,8 --sampling '{"temperature": 0.8, "top_k": 20, "top_p": 0.9}'
run sampling_minp --job 2,2,16,8,8 --sampling '{"temperature": 1.3, "min_p": 0.05}'
