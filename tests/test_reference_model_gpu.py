"""The plug-in under the REAL reference model stack on MI355X (VERDICT r03 missing #1 / #3).

Nothing from sglang_amd/harness and no stand-in: the process (tests/golden/ref_model.py --run gpu, one per model size)
  * lets the reference's `load_plugins()` (srt/plugins/__init__.py:103-141) discover the package through its entry points
    and execute `plugin.load()`; `current_platform` (srt/platforms/__init__.py) resolves to the package's out-of-tree platform;
  * publishes a `ServerArgs` with the reference's defaults + `attention_backend = current_platform.get_default_attention_backend()`,
    initialises the reference's distributed state (RCCL, one rank) and builds the reference's `LlamaForCausalLM` on cuda:0
    from a Hugging-Face-named checkpoint through the reference's own `load_weights`;
  * builds the attention backend from the reference's registry, and runs cold extend / warm extend over a cached prefix /
    decode steps as ModelRunner does (`ForwardBatch.init_new` -> `init_forward_metadata` -> `model.forward` inside
    `forward_context`): prefill through the reference's layer loop with the registered operator forwards, the hooked
    `UnquantizedLinearMethod.apply` and the hooked position functions; decode through the AROUND hook on
    `LlamaModel.forward` (the fused 9-launch layer) -- eagerly and inside a hipGraph captured with the reference's protocol;
  * samples with `create_sampler(<backend>)` and compares the seeded top-k / top-p ids with the reference's own `Sampler`.
Every pass is compared with oracle/model.py (bit-identical to the reference's torch-native forward: tests/test_reference_model.py).
The test skips when no reference sources are staged (python tests/golden/ref_model.py --run stage in the build container).
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests" / "golden"))


@pytest.mark.parametrize("dims,tp", [("tiny", 1), ("llama3_8b_2layers", 1), ("tiny_qwen2", 1), ("tiny", 2), ("tiny_mixtral", 1)])
def test_plugin_under_the_references_model_stack(device, dims, tp):
    """tp = 2: two processes on GPU 0 under the reference's `initialize_model_parallel(2)` (gloo device groups: RCCL refuses
    two ranks of one device) -- the reference's GroupCoordinator constructor attaches the xGMI communicator through the
    plug-in's hook, the row-parallel projections and the vocab-parallel embedding all-reduce through the hooked
    `GroupCoordinator.all_reduce`, the logits gather through the hooked `all_gather`, and the decode passes run the fused
    layer with the all-reduce + add + RMSNorm epilogue."""
    import ref_model

    if ref_model.ref_root() is None:
        pytest.skip("reference sources are not staged (python tests/golden/ref_model.py --run stage in the build container)")
    out = ROOT / "gpurun_out" / f"reference_model_{dims}{'_tp%d' % tp if tp > 1 else ''}.json"
    env = dict(os.environ, SGLANG_USE_AITER="0")                     # (aiter is not in this image; the reference's default is off)
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "golden" / "ref_model.py"), "--run", "gpu", "--dims", dims, "--tp", str(tp),
                        "--json", str(out)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-6000:]
    rep = json.loads(out.read_text())
    ld = rep["loader"]
    assert ld["platform"] == "Mi355xSRTPlatform" and ld["out_of_tree"] and ld["attention_backend_registered"] and ld["sampler_registered"]
    assert len(ld["hooks_applied"]) == len(ld["hooked"]) == 13      # model x3, TP x4, linear + lm_head, positions x2, allocation x2
    assert rep["unstaged_reference_modules"] == [], rep["unstaged_reference_modules"]
    for name in ("build", "prefill", "decode", "sampler", "graph_decode"):
        assert rep["legs"].get(name, {}).get("ok"), (name, rep["legs"].get(name))
    b = rep["legs"]["build"]
    model_cls = {"tiny_qwen2": "Qwen2ForCausalLM", "tiny_mixtral": "MixtralForCausalLM"}.get(dims, "LlamaForCausalLM")
    assert (b["backend"], b["model"], b["rope"]) == ("HipAttnBackend", model_cls, "RotaryEmbedding")
    moe = dims == "tiny_mixtral"
    assert rep["tp"] == tp
    if tp > 1:
        c = rep["counts"]
        # per layer of a prefill pass: o_proj + down_proj all-reduces (+ the embedding's); per fused decode layer: two all-reduces
        # with the add + RMSNorm epilogue; every forward gathers the logits
        assert c["xgmi_attached"] and c["xgmi_all_reduce"] >= 10 and c["xgmi_all_reduce_add_rmsnorm"] >= 20 and c["xgmi_all_gather"] >= 7, c
    # the decode passes ran the fused layer loop: 3 eager steps + the graph's warm-up and capture; the prefill passes did not
    assert rep["counts"]["fused_decode_models"] == 5, rep["counts"]
    if moe:
        # MixtralModel.forward carries the model-level hook too (round 5): decode passes run the fused attention half and call the
        # block's own gate / TopK / experts; every pass' experts -- prefill through the reference's layer loop, decode through the
        # hook -- go UnquantizedFusedMoEMethod's registered forward -> MoeRunner.run -> the function in FusedOpPool's slot -> the
        # gfx950 grouped GEMMs: 7 forwards x 2 layers, none handed back to the reference's Triton function
        assert rep["counts"]["moe_fused_func_calls"] == 14 and rep["counts"]["moe_hip_calls"] == 14, rep["counts"]
    # the hooked UnquantizedLinearMethod.apply saw the prefill projections: 187 rows -> the library GEMM, the 50-row warm extend
    # -> the weight stream (4 projections x layers each); lm_head rows (3, 1) stream as well
    assert rep["counts"]["library_linears"] >= 8 and rep["counts"]["streamed_linears"] >= 8, rep["counts"]
    # every fused-op call of the prefill passes was served by a forward plugin.load() registered (label = the method name,
    # fused_op.py `_dispatch_label`), none by the reference's torch / hip / triton forwards
    tr = rep["fused_op_trace"]
    want_ops = ["RMSNorm:forward", "RotaryEmbedding:forward", "TopK:forward", "UnquantizedFusedMoEMethod:cuda"] if moe else [
        "RMSNorm:forward", "RotaryEmbedding:forward", "SiluAndMul:forward"]
    assert sorted(tr) == want_ops, tr
    assert len(rep["passes"]) == 7
    for ps in rep["passes"]:
        # Two bf16 evaluations of a 2-layer model differ by a few logit ulps (the reference's literal evaluation vs its own
        # fp32-accumulating one does): the plug-in's error against the fp32-accumulating oracle must stay inside the band of the
        # reference's literal evaluation against the same oracle -- rms within 1.25x, the single worst logit within 2x -- and
        # every clear-margin arg-max must agree with the literal evaluation.
        if moe:
            # (routing is discrete: a token whose 2nd / 3rd router scores tie within bf16 noise may take another expert in either
            # evaluation, which moves that row by far more than rounding does -- the rms band is widened, single logits are not judged)
            assert ps["product_rms_err"] <= 2.0 * ps["reference_rms_err"] + 1e-3 and ps["argmax_agree"] >= ps["clear_rows"] - 1, ps
            continue
        assert ps["product_rms_err"] <= 1.25 * ps["reference_rms_err"] + 1e-4, ps
        assert ps["product_max_err"] <= 2.0 * ps["reference_max_err"] + 1e-3, ps
        assert ps["argmax_agree"] == ps["clear_rows"] and ps["max_ulp"] <= 8.0, ps
    s = rep["legs"]["sampler"]
    assert s["is_reference_subclass"] and s["greedy_equals_argmax"] and s["seeded_ids_equal"], s


def test_plugin_under_the_references_model_runner_with_an_fp8_kv_pool(device):
    """`--kv-cache-dtype fp8_e4m3` under the reference's ModelRunner: the reference's pool holds float8_e4m3fn rows, the plug-in's store /
    attention kernels and the fused decode layer write and read them (captured in the reference's decode graphs); logits inside the band
    of the reference's literal evaluation over the same e4m3 rows."""
    import ref_model

    if ref_model.ref_root() is None:
        pytest.skip("reference sources are not staged (python tests/golden/ref_model.py --run stage in the build container)")
    out = ROOT / "gpurun_out" / "reference_model_runner_fp8kv.json"
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "golden" / "ref_model.py"), "--run", "runner", "--server-args",
                        '{"kv_cache_dtype": "fp8_e4m3"}', "--json", str(out)],
                       cwd=ROOT, env=dict(os.environ, SGLANG_USE_AITER="0"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-6000:]
    rep = json.loads(out.read_text())
    assert rep["kv_pool_dtype"] == "torch.float8_e4m3fn" and rep["attn_backend_class"] == "HipAttnBackend"
    c = rep["counts"]
    assert rep["fused_decode_models_during_capture"] > 0 and (c["graph_replays"], c["not_fused_because"]) == (4, []), c
    for ps in rep["passes"]:
        assert ps["product_rms_err"] <= 1.25 * ps["reference_rms_err"] + 1e-3, ps
        assert ps["product_max_err"] <= 2.0 * ps["reference_max_err"] + 1e-2, ps


@pytest.mark.parametrize("dims,model", [("tiny", "LlamaForCausalLM"), ("tiny_qwen2", "Qwen2ForCausalLM"), ("tiny_mixtral", "MixtralForCausalLM")])
def test_plugin_under_the_references_model_runner(device, dims, model):
    """The reference's `ModelRunner` itself, on MI355X with the plug-in (tests/golden/ref_model.py run_runner; its CPU twin:
    tests/test_reference_model.py): `ServerArgs(attention_backend=None)` resolves the backend name from the out-of-tree platform,
    `ModelRunner` builds the model, the pools (the platform's `get_mha_kv_pool_cls`), the backend (the registry's factory) and the
    decode graphs (the platform's `get_graph_runner_cls` -> the reference's DecodeCudaGraphRunner capturing the hooked model at
    every batch size), and the reference's static-batch harness `sglang.benchmark.one_batch` drives prefill / extend-over-prefix /
    decode with real `Req` / `ScheduleBatch` objects; decode forwards are graph replays, sampling goes through `ModelRunner.sample`."""
    import ref_model

    if ref_model.ref_root() is None:
        pytest.skip("reference sources are not staged (python tests/golden/ref_model.py --run stage in the build container)")
    out = ROOT / "gpurun_out" / f"reference_model_runner{'' if dims == 'tiny' else '_' + dims}.json"
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "golden" / "ref_model.py"), "--run", "runner", "--dims", dims, "--json", str(out)],
                       cwd=ROOT, env=dict(os.environ, SGLANG_USE_AITER="0"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-6000:]
    rep = json.loads(out.read_text())
    assert rep["loader"]["out_of_tree"] and rep["attention_backend"] == "hip_mi355x"          # resolved by ServerArgs from the platform
    assert (rep["attn_backend_class"], rep["sampler_class"], rep["model"], rep["kv_pool"]) == ("HipAttnBackend", "HipSampler", model, "Mi355xMHATokenToKVPool")
    assert rep["graph_runner"] == "DecodeCudaGraphRunner" and rep["captured_batch_sizes"], rep
    c = rep["counts"]
    moe = model == "MixtralForCausalLM"
    # every capture ran the fused decode layer loop (Mixtral's sparse-MoE form included, round 5); the four decode steps were graph
    # replays (no further eager decode forward)
    assert rep["fused_decode_models_during_capture"] >= len(rep["captured_batch_sizes"])
    assert (c["fused_decode_models"], c["graph_replays"], c["not_fused_because"]) == (rep["fused_decode_models_during_capture"], 4, []), c
    assert [u for u in rep["unstaged_reference_modules"] if not u.startswith("sglang._version")] == []
    assert len(rep["passes"]) == 6
    for ps in rep["passes"]:
        if moe:                         # (discrete routing: see test_plugin_under_the_references_model_stack)
            assert ps["product_rms_err"] <= 2.0 * ps["reference_rms_err"] + 1e-3, ps
            continue
        assert ps["product_rms_err"] <= 1.25 * ps["reference_rms_err"] + 1e-4, ps
        assert ps["product_max_err"] <= 2.0 * ps["reference_max_err"] + 1e-3, ps


@pytest.mark.parametrize("loop", ["normal", "overlap", "overlap-paged-chunked", "overlap-logprobs", "overlap-mixtral", "overlap-mixed", "overlap-retract"])
def test_plugin_under_the_references_scheduler(device, loop):
    """The reference's `Scheduler` itself on MI355X with the plug-in, running its own `run_event_loop()` (tests/golden/ref_model.py
    run_scheduler_job; CPU twin in tests/test_reference_model.py): intake, prefill admission, radix cache, continuous batching,
    `TpModelWorker` -> `ModelRunner` -> the captured decode graphs, result processing and output streaming -- `event_loop_normal`
    and the server's default `event_loop_overlap` (forward of batch N launched before the results of batch N-1 are processed).  The
    later requests hit the shared tokens in the reference's radix tree; the plug-in's kernels produce the tokens; decode batches
    are replays of the reference's graphs.  `overlap-paged-chunked`: `--page-size 16 --chunked-prefill-size 64` on top (the reference's
    paged allocator and page-aligned radix keys; 104-token prompts prefilled in chunks that extend over the request's own earlier
    chunks, mixed with the running decode batches)."""
    import ref_model

    if ref_model.ref_root() is None:
        pytest.skip("reference sources are not staged (python tests/golden/ref_model.py --run stage in the build container)")
    out = ROOT / "gpurun_out" / f"reference_model_scheduler_{loop}.json"
    variant = loop == "overlap-paged-chunked"
    extra = ["--overlap"] if loop != "normal" else []
    if variant:
        extra += ["--job", "2,3,80,24,6", "--server-args", '{"page_size": 16, "chunked_prefill_size": 64}']
    if loop == "overlap-retract":       # a KV pool too small for the job, admitted aggressively: the scheduler retracts and re-prefills
        extra += ["--job", "2,4,32,16,40", "--server-args", '{"max_total_tokens": 380, "schedule_conservativeness": 0.05}']
    if loop == "overlap-mixed":         # --enable-mixed-chunk: prefill chunks and running decodes in one ForwardMode.MIXED forward
        extra += ["--job", "2,3,80,24,6", "--server-args", '{"chunked_prefill_size": 64, "enable_mixed_chunk": true}']
    if loop == "overlap-logprobs":      # every request with return_logprob + top-2: the plug-in sampler's log-probability outputs
        extra += ["--logprobs"]
    if loop == "overlap-mixtral":       # MixtralForCausalLM: no model-level hook; hooked projections, registered ops, the MoE slot
        extra += ["--dims", "tiny_mixtral"]
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "golden" / "ref_model.py"), "--run", "scheduler", "--json", str(out)] + extra,
                       cwd=ROOT, env=dict(os.environ, SGLANG_USE_AITER="0"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-6000:]
    rep = json.loads(out.read_text())
    assert (rep["scheduler"], rep["tp_worker"], rep["attention_backend"], rep["attn_backend_class"], rep["sampler_class"], rep["graph_runner"],
            rep["event_loop"]) == ("Scheduler", "TpModelWorker", "hip_mi355x", "HipAttnBackend", "HipSampler", "DecodeCudaGraphRunner",
                                   "normal" if loop == "normal" else "overlap")
    for job in (rep["warm_up"], rep["timed"]):
        b = job["batches_run"]
        if loop == "overlap-retract":
            assert job["finished_requests"] == 8 and job["tokens_per_request"] == [40] and job["cached_tokens_of_leaders"] == [0]
        elif loop == "overlap-mixed":
            assert sum(v for k, v in b.items() if k.startswith("MIXED")) >= 2, b
            assert job["cached_tokens_of_others"] == [80] and job["finished_requests"] == 6 and job["tokens_per_request"] == [6]
        elif variant:
            assert (rep["page_size"], rep["chunked_prefill_size"]) == (16, 64)
            assert sum(v for k, v in b.items() if k.startswith("EXTEND")) >= 4 and b.get("DECODE x6", 0) >= 1, b
            assert job["cached_tokens_of_leaders"] == [0] and job["cached_tokens_of_others"] == [80]
            assert job["finished_requests"] == 6 and job["tokens_per_request"] == [6]
        else:
            assert b["EXTEND x2"] == 2 and b.get("DECODE x4", 0) >= 1 and sum(v for k, v in b.items() if k.startswith("DECODE")) >= 3, b
            assert job["cached_tokens_of_leaders"] == [0] and job["cached_tokens_of_others"] == [16]
            assert job["finished_requests"] == 4 and job["tokens_per_request"] == [4]
    moe = loop == "overlap-mixtral"
    assert rep["fused_decode_models_during_capture"] > 0 and rep["eager_fused_decode_forwards_in_the_timed_job"] == 0
    assert rep["graph_replays_in_the_timed_job"] >= 3
    # ---- nothing of the job runs on a Triton kernel (north_star: "no Triton dispatch"): every `kernel[grid](...)` of the process goes
    # through a counting `JITFunction.run`; the reference's pool / allocator objects are the platform factories' subclasses, its
    # allocation helpers the hooked functions, and no norm call was handed to the reference's torch forward
    assert rep["triton_launches_in_the_timed_job"] == 0, rep["triton_kernels_in_the_timed_job"]
    assert (rep["kv_pool_class"], rep["allocator_class"]) == ("Mi355xMHATokenToKVPool", "Mi355xPagedTokenToKVPoolAllocator")
    pc = rep["plugin_counts"]
    assert pc["rmsnorm"]["native"] == 0 and pc["rmsnorm"]["hip"] > 0, pc
    assert pc["mem_hooks"]["write_cache_indices"] >= 2 and pc["mem_hooks"]["store_kv"] >= 4, pc
    if variant:        # page size 16: the paged allocator's two kernels and the last-slot lookup are the gfx950 ones
        assert pc["mem_hooks"]["alloc_extend"] >= 2 and pc["mem_hooks"]["alloc_decode"] >= 2, pc
    # ---- every forward's logits against the oracle, teacher-forced with the produced tokens: the plug-in's error
    # against the fp32-accumulating oracle inside the band of the reference's literal bf16 evaluation against the same oracle
    lb = rep["logit_band"]
    want_rows = {"overlap-retract": 320, "overlap-mixed": 36, "overlap-paged-chunked": 36}.get(loop, 16)
    assert lb["rows_compared"] == lb["rows_expected"] == want_rows, lb
    if moe:            # (discrete routing: a flipped expert moves a whole row -- see test_plugin_under_the_references_model_stack)
        assert lb["product_rms_err"] <= 2.0 * lb["reference_rms_err"] + 1e-3, lb
    else:
        # (MIXED / chunked forwards evaluate a token through another reduction structure than the oracle's one-shot prefill + decode:
        # the reference's own backend does too -- CPU twin: a couple of logit ulps -- so those two get 1.5x)
        slack = 1.5 if loop in ("overlap-mixed", "overlap-paged-chunked") else 1.25
        assert lb["product_rms_err"] <= slack * lb["reference_rms_err"] + 1e-4, lb
        assert lb["product_max_err"] <= 2.0 * lb["reference_max_err"] + 1e-3, lb
        assert lb["argmax_agree_on_clear_rows"] >= lb["clear_rows"] - (1 if slack > 1.25 else 0), lb
    if loop == "overlap-retract":
        assert rep["retracted_requests"] >= 1 and rep["max_total_num_tokens"] == 380, rep["retracted_requests"]
    if loop == "overlap-mixed":         # (a decode token inside a MIXED forward takes the extend path: another reduction order)
        assert rep["oracle"]["token_agreement"] >= 0.6, rep["oracle"]
        return
    # greedy tokens of a random-weight model: a near-tie may flip between two bf16 evaluations (for the MoE model a flipped expert
    # choice too); most tokens must agree exactly
    o = rep["oracle"]
    # (40 free-running greedy tokens per request in the retraction case: once a near-tie flips, the rest of that request differs)
    assert o["token_agreement"] >= (0.6 if moe or loop == "overlap-retract" else 0.75), o
    if loop == "overlap-logprobs":
        # the streamed log-probabilities against log_softmax of the oracle's logits (teacher-forced with the produced tokens): bf16
        # logits of rms 1 carry ~0.01-0.03 of evaluation noise
        assert o["logprob_values"] == 16 and o["max_abs_logprob_diff"] <= 0.08 and o["top2_sets_equal"] >= 12, o


def test_mem_hooks_equal_the_references_own_device_code(device):
    """Round 5's take-overs of the scheduler's slot bookkeeping and KV store, differentially against the REFERENCE'S OWN device code
    on the same inputs (tests/golden/ref_model.py run_mem_hooks; the reference's Triton kernels run on MI355X): the platform's paged
    allocator against `PagedTokenToKVPoolAllocator` through random request histories at page sizes 1 / 4 / 16 (outputs AND free lists
    after every call; int32 `last_loc` as `alloc_for_decode` reads it out of the request table), the hooked `write_cache_indices` /
    `get_last_loc` against `write_req_to_token_pool_triton` / `get_last_loc_triton_safe`, the platform's pool class against
    `MHATokenToKVPool.set_kv_buffer` for bf16 and e4m3 pools -- all bit-exact, none of ours launching Triton."""
    import ref_model

    if ref_model.ref_root() is None:
        pytest.skip("reference sources are not staged (python tests/golden/ref_model.py --run stage in the build container)")
    out = ROOT / "gpurun_out" / "reference_mem_hooks.json"
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "golden" / "ref_model.py"), "--run", "mem-hooks", "--json", str(out)],
                       cwd=ROOT, env=dict(os.environ, SGLANG_USE_AITER="0"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-6000:]
    rep = json.loads(out.read_text())
    assert [a["page_size"] for a in rep["allocator"]] == [1, 4, 16] and all(a["equal"] and a["calls"] >= 6 for a in rep["allocator"]), rep["allocator"]
    for key in ("write_cache_indices", "get_last_loc"):
        assert len(rep[key]) == 3 and all(r["equal"] and r["served_by_the_hook"] and r["triton_launches"] == 0 for r in rep[key]), rep[key]
    assert len(rep["kv_store"]) == 2 and all(r["equal"] and r["served_by_the_pool_class"] and r["nonzero"] for r in rep["kv_store"]), rep["kv_store"]


def test_target_verify_under_the_references_scheduler(device):
    """`--speculative-algorithm NGRAM --speculative-num-draft-tokens 4` under the reference's `Scheduler` (overlap loop) with the plug-in
    (VERDICT r04 missing #5): the reference's `NGRAMWorker` turns every decode batch into a `ForwardMode.TARGET_VERIFY` forward of four
    draft tokens per request under a tree mask (`NgramVerifyInput.custom_mask`), its graph runner CAPTURES that mode around the hooked
    model, `eagle_sample` accepts the drafts the target agrees with, the KV mover compacts the accepted rows.  What is not in this
    image -- the C++ n-gram corpus and two sgl_kernel ops -- are test stand-ins (tests/golden/ref_model.py `_install_spec_standins`:
    a scripted drafter proposing the oracle's greedy continuation with one wrong token per draft; the reference's own Triton form of
    the greedy tree walk).  Speculative decoding must not change greedy output: the tokens are compared with the oracle's greedy
    generation, and the job must take fewer verify forwards than tokens (drafts WERE accepted)."""
    import ref_model

    if ref_model.ref_root() is None:
        pytest.skip("reference sources are not staged (python tests/golden/ref_model.py --run stage in the build container)")
    out = ROOT / "gpurun_out" / "reference_model_scheduler_target_verify.json"
    n_out = 12
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "golden" / "ref_model.py"), "--run", "scheduler", "--overlap", "--spec-ngram", "4",
                        "--spec-tree", "--job", f"2,2,16,8,{n_out}", "--json", str(out)],
                       cwd=ROOT, env=dict(os.environ, SGLANG_USE_AITER="0"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-6000:]
    rep = json.loads(out.read_text())
    sp = rep["spec"]
    assert (rep["scheduler"], rep["attn_backend_class"], sp["worker"], sp["draft_tokens"]) == ("Scheduler", "HipAttnBackend", "NGRAMWorker", 4), sp
    modes = sp["forward_modes_in_the_timed_job"]
    verify = modes.get("TARGET_VERIFY", 0)
    assert verify >= 3 and modes.get("EXTEND", 0) >= 2, modes
    # every (request, verify step) is one drafter lookup: without accepted drafts the four requests need 4 x 11 of them
    assert sp["drafter"]["lookups"] < 4 * (n_out - 1), ("no draft was ever accepted", sp["drafter"])
    assert sp["drafter"]["matched"] > 0 and sp["drafter"]["drafted_true_tokens"] > 0, sp["drafter"]
    for job in (rep["warm_up"], rep["timed"]):
        assert job["finished_requests"] == 4 and job["tokens_per_request"] == [n_out], job
        assert job["cached_tokens_of_others"] == [16]
    assert rep["graph_replays_in_the_timed_job"] >= 3                              # the verify forwards are replays of TARGET_VERIFY graphs
    assert rep["oracle"]["token_agreement"] >= 0.75, rep["oracle"]
    # the drafter proposed chains AND trees with wrong branches (a wrong first sibling, a rejected second sibling): both walked
    assert sp["drafter"]["tree_drafts"] >= 2 and sp["drafter"]["chain_drafts"] >= 1, sp["drafter"]
    _spec_logit_bars(sp, dense=True)


def _spec_logit_bars(sp, dense: bool) -> None:
    """every row of every TARGET_VERIFY forward -- accepted and rejected draft nodes alike -- against the oracle's
    evaluation of that node's own token path (tests/golden/ref_model.py spec_logit_band), with the scheduler tests' bars: the plug-in's
    error against the fp32-accumulating oracle inside 1.25 x (rms) / 2 x (worst logit) the band of the reference's literal bf16
    evaluation; a wrong-but-close custom-mask kernel, a stale tree mask or a wrong position fails here, not at token agreement."""
    lb = sp["logit_band"]
    assert lb["rows_compared"] == lb["rows_expected"] > 0 and lb["verify_forwards"] >= 3, lb
    assert len(lb["rows_by_depth"]) >= 3, lb                          # roots, children, grandchildren were all scored
    if dense:
        assert lb["product_rms_err"] <= 1.25 * lb["reference_rms_err"] + 1e-4, lb
        assert lb["product_max_err"] <= 2.0 * lb["reference_max_err"] + 1e-3, lb
        assert lb["argmax_agree_on_clear_rows"] == lb["clear_rows"], lb
    else:
        # sparse-MoE: a flipped expert choice moves a whole row; rows whose token path holds a near-tie of the router (or on which the
        # two oracles themselves route differently) keep the 2 x rms bar only, the others get the per-logit bar as well
        assert lb["product_rms_err"] <= 2.0 * lb["reference_rms_err"] + 1e-3, lb
        assert lb["rows_without_flip_risk"] >= lb["rows_compared"] // 4, lb
        assert lb["product_rms_err_no_flip"] <= 1.25 * lb["reference_rms_err_no_flip"] + 1e-4, lb
        assert lb["product_max_err_no_flip"] <= 2.0 * lb["reference_max_err_no_flip"] + 1e-3, lb


@pytest.mark.parametrize("config", ["eager", "small-graph", "fp8-kv", "qwen2", "no-radix", "tp2", "spec-mixtral", "spec-paged", "long-shared",
                                    "long-shared-paged", "sampling", "sampling-wide-vocab"])
def test_plugin_under_the_references_scheduler_more_configurations(device, config):
    """Configurations of the reference's scheduler (overlap loop) beyond the seven of test_plugin_under_the_references_scheduler, added
    in round 5 after two of them exposed defects: no decode graphs at all; graphs for 2 requests with 6 running (replays and eager
    decode forwards of the hooked model interleaved); an fp8 e4m3 KV pool (oracle over e4m3 rows too); Qwen2 (qkv bias, tied
    embeddings); `--disable-radix-cache` (ChunkCache: requests carry EMPTY host-side prefix tensors -- the allocation hook declined
    them and the Triton writer ran); TP = 2 (two processes, the reference's GroupCoordinator + the xGMI communicator under the
    scheduler); NGRAM speculative decoding on the sparse-MoE model and at page size 16 (chain drafts); groups of four requests
    sharing 160 tokens over 48 decode steps at page sizes 1 and 16 (the shared-prefix decode plan finds a >= 128-token group under
    the reference's tables, contexts cross chunk boundaries; 384 logit rows in the band); every request SAMPLING (temperature 0.8,
    top-k 20, top-p 0.9 through the scheduler's SamplingBatchInfo: each delivered token lies in the top-k set of the logits row it was
    sampled from, the rows -- teacher-forced with the sampled tokens -- in the band); the same at Llama-3-8B's width and vocabulary
    (two layers), where the scheduler's fp32 [4, 128256] logits take the one-call sampler that never writes the probabilities
    (round 6)."""
    import ref_model

    if ref_model.ref_root() is None:
        pytest.skip("reference sources are not staged (python tests/golden/ref_model.py --run stage in the build container)")
    extra = {"eager": ["--server-args", '{"disable_cuda_graph": true}'],
             "small-graph": ["--job", "2,3,16,8,6", "--server-args", '{"cuda_graph_max_bs_decode": 2}'],
             "fp8-kv": ["--server-args", '{"kv_cache_dtype": "fp8_e4m3"}'],
             "qwen2": ["--dims", "tiny_qwen2"],
             "no-radix": ["--server-args", '{"disable_radix_cache": true}'],
             "tp2": ["--tp", "2"],
             "spec-mixtral": ["--spec-ngram", "3", "--spec-tree", "--dims", "tiny_mixtral", "--job", "2,2,16,8,10"],
             "spec-paged": ["--spec-ngram", "4", "--job", "2,2,32,16,12", "--server-args", '{"page_size": 16, "speculative_ngram_max_bfs_breadth": 1}'],
             "long-shared": ["--job", "2,4,160,40,48"],
             "long-shared-paged": ["--job", "2,4,160,40,48", "--server-args", '{"page_size": 16}'],
             "sampling": ["--job", "2,2,16,8,8", "--sampling", '{"temperature": 0.8, "top_k": 20, "top_p": 0.9}'],
             "sampling-wide-vocab": ["--dims", "llama3_8b_2layers", "--job", "2,2,16,8,8", "--sampling",
                                     '{"temperature": 0.8, "top_k": 20, "top_p": 0.9}']}[config]
    out = ROOT / "gpurun_out" / f"reference_model_scheduler_more_{config}.json"
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "golden" / "ref_model.py"), "--run", "scheduler", "--overlap", "--json", str(out)] + extra,
                       cwd=ROOT, env=dict(os.environ, SGLANG_USE_AITER="0"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-6000:]
    rep = json.loads(out.read_text())
    spec = config.startswith("spec")
    assert rep["attn_backend_class"] == "HipAttnBackend" and rep["kv_pool_class"] == "Mi355xMHATokenToKVPool"
    n_req, n_out = {"small-graph": (6, 6), "spec-mixtral": (4, 10), "spec-paged": (4, 12), "long-shared": (8, 48), "long-shared-paged": (8, 48),
                    "sampling": (4, 8), "sampling-wide-vocab": (4, 8)}.get(config, (4, 4))
    for job in (rep["warm_up"], rep["timed"]):
        assert job["finished_requests"] == n_req and job["tokens_per_request"] == [n_out], job
    if config == "eager":
        assert rep["graph_replays_in_the_timed_job"] == 0 and rep["eager_fused_decode_forwards_in_the_timed_job"] >= 3, rep     # the fused layer, eagerly
    elif config == "small-graph":
        assert rep["graph_replays_in_the_timed_job"] >= 1 and rep["eager_fused_decode_forwards_in_the_timed_job"] >= 2, rep
    else:
        assert rep["graph_replays_in_the_timed_job"] >= 3, rep
    if spec:
        sp = rep["spec"]
        verify = sp["forward_modes_in_the_timed_job"].get("TARGET_VERIFY", 0)
        # every (request, verify step) is one drafter lookup; without accepted drafts a request needs n_out - 1 of them
        assert sp["worker"] == "NGRAMWorker" and verify >= 3 and sp["drafter"]["drafted_true_tokens"] > 0, sp
        assert sp["drafter"]["lookups"] < n_req * (n_out - 1), ("no draft was ever accepted", sp["drafter"])
        assert rep["oracle"]["token_agreement"] >= 0.75, rep["oracle"]
        _spec_logit_bars(sp, dense=config != "spec-mixtral")
        return
    # no Triton launch on the path; the logits of every forward inside the reference's own band
    assert rep["triton_launches_in_the_timed_job"] == 0, rep["triton_kernels_in_the_timed_job"]
    lb = rep["logit_band"]
    assert lb["rows_compared"] == lb["rows_expected"] == n_req * n_out, lb
    assert lb["product_rms_err"] <= 1.25 * lb["reference_rms_err"] + 1e-4, lb
    assert lb["product_max_err"] <= 2.0 * lb["reference_max_err"] + 1e-3, lb
    assert lb["argmax_agree_on_clear_rows"] == lb["clear_rows"], lb
    if config.startswith("sampling"):
        sm = rep["sampling"]
        assert sm["tokens_checked"] == sm["tokens_inside_their_rows_top_k"] == n_req * n_out and sm["distinct_first_tokens"] >= 2, sm
        routes = rep["plugin_counts"]["sampler"]
        if config == "sampling-wide-vocab":
            assert routes["one_call_from_logits"] >= n_out - 1 and routes["softmax_then_sample"] == 0, routes
        return
    if config.startswith("long-shared"):
        assert rep["timed"]["cached_tokens_of_others"] == [160]
        # the backend's plan found the two groups in the reference's request table: their 128-token shared chunk is read once per group
        assert rep["last_decode_plan"]["groups"] == 2 and rep["last_decode_plan"]["shared_items"] >= 2, rep["last_decode_plan"]
        # (48 free-running greedy tokens of a random-weight model: after a near-tie flips the rest of that request differs -- the
        # teacher-forced logit band above is the bar; most tokens still agree)
        assert rep["oracle"]["token_agreement"] >= 0.5, rep["oracle"]
        return
    assert rep["oracle"]["token_agreement"] >= 0.75, rep["oracle"]
