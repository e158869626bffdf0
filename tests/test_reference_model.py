"""The oracle's whole-model restatement pinned on the REAL reference model, and the plug-in loaded by the REAL loader --
both in the build container, without a GPU (VERDICT r03 missing #1, weak #1c).

tests/golden/ref_model.py imports the reference's own `LlamaForCausalLM`, `ServerArgs`, `init_distributed_environment`,
`ForwardBatch`, pools, `LogitsProcessor`, `Sampler`, `load_plugins` ... from /root/reference (or from the staged copy under
oracle/_ref/sglang_model) through the import hook of gen_golden.py.  Each run is its own process: the hook, the global server
args, the process group and the hooks on the reference's classes do not leak into the other tests.

  cpu-oracle   `LlamaForCausalLM.load_weights(HF-named checkpoint)` -> cold extend, warm extend over a cached prefix, three
               decode steps with the reference's torch-native attention backend and torch operator forwards, against
               oracle/model.py on the reference model's own (loader-stacked) parameters: the logits must be IDENTICAL.
  loader       `sglang.srt.plugins.load_plugins()` finds `sglang_amd.plugin:load` through the entry points of a dist-info
               directory equal to what `pip install -e .` writes, executes it against the real registries and applies the
               hooks to the real classes; `sglang.srt.platforms.current_platform` resolves to the package's platform.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests" / "golden"))


def _run(mode, tmp_path, env=None, timeout=900, extra=()):
    out = tmp_path / f"{mode}.json"
    e = dict(os.environ, **(env or {}))
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "golden" / "ref_model.py"), "--run", mode, "--json", str(out), *extra],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    return json.loads(out.read_text())


def _root_or_skip():
    import ref_model

    root = ref_model.ref_root()
    if root is None:
        pytest.skip("no reference sources here (neither /root/reference nor a staged copy)")
    return root


def test_oracle_whole_model_equals_the_references_llama_forward(tmp_path):
    _root_or_skip()
    rep = _run("cpu-oracle", tmp_path)
    assert [p["what"] for p in rep["passes"]] == ["cold extend 37+130+20", "warm extend 50 over a 64-token prefix"] + [
        f"decode step {i} (4 requests)" for i in range(3)]
    for p in rep["passes"]:
        assert p["identical"] and p["max_abs"] == 0.0 and p["ref_rms"] > 0.5, p          # bit-identical logits, non-trivial ones
    s = rep["sampler"]
    assert s["sampler_class"] == "Sampler" and s["is_reference_subclass"] and s["greedy_equals_argmax"]


@pytest.mark.parametrize("extra,what", [(("--dims", "qwen2.5_0.5b"), "BASELINE configs[0]'s architecture, whole depth, on Qwen2ForCausalLM"),
                                        (("--tp", "2"), "two ranks: the reference's groups, row-parallel all-reduces, vocab-parallel embedding + logits all-gather"),
                                        (("--dims", "tiny_mixtral"), "MixtralForCausalLM: router -> TopK -> FusedMoE's native forward (fused_moe_native.py)")])
def test_oracle_equals_the_reference_on_qwen2_and_at_tp2(tmp_path, extra, what):
    """The same five passes: (a) the reference's `Qwen2ForCausalLM` at Qwen2.5-0.5B's own dimensions (24 layers, qkv bias, tied
    embeddings: the configuration BASELINE.json runs on the CPU), (b) `LlamaForCausalLM` under `initialize_model_parallel(2)` --
    two gloo processes, every rank's shards cut by the reference's own weight loader -- against the oracle's TP = 2 mode."""
    _root_or_skip()
    rep = _run("cpu-oracle", tmp_path, extra=extra)
    assert len(rep["passes"]) == 5 and all(p["identical"] and p["ref_rms"] > 0.5 for p in rep["passes"]), (what, rep["passes"])
    assert rep["tp"] == (2 if "--tp" in extra else 1)


def test_oracle_fp8_kv_mode_equals_the_references_fp8_pool(tmp_path):
    """`--kv-cache-dtype fp8_e4m3`: the reference's `MHATokenToKVPool` stores float8_e4m3fn rows (memory_pool.py:2364-2374) and its
    torch-native backend reads them back; the oracle's `kv_cache_dtype="fp8_e4m3"` mode reproduces every logit bit for bit."""
    _root_or_skip()
    rep = _run("runner", tmp_path, extra=("--server-args", '{"kv_cache_dtype": "fp8_e4m3"}'))
    assert rep["kv_pool_dtype"] == "torch.float8_e4m3fn"
    assert len(rep["passes"]) == 6 and all(p["identical"] and p["ref_rms"] > 0.5 for p in rep["passes"]), rep["passes"]


@pytest.mark.parametrize("dims,model", [("tiny", "LlamaForCausalLM"), ("tiny_qwen2", "Qwen2ForCausalLM"), ("tiny_mixtral", "MixtralForCausalLM")])
def test_oracle_equals_the_reference_under_the_references_model_runner(tmp_path, dims, model):
    """One level up (tests/golden/ref_model.py run_runner): the reference's real `ServerArgs` (its whole resolution pipeline),
    `ModelConfig.from_server_args`, `ModelRunner` (distributed init, model loader, KV-cache configurator, pools, allocator, attention
    backend from the registry) driven by the reference's own static-batch harness `sglang.benchmark.one_batch` -- real `Req` /
    `ScheduleBatch.prepare_for_extend / prepare_for_decode` (the reference's allocator picks the slots), `ForwardBatch.init_new`,
    `ModelRunner.forward`, `ModelRunner.sample` -- following its correctness test: prefill of a cut, extend over the cached prefix,
    greedy decode.  On the CPU with the torch-native backend the oracle reproduces every logit bit for bit."""
    _root_or_skip()
    rep = _run("runner", tmp_path, extra=("--dims", dims))
    assert (rep["device"], rep["attn_backend_class"], rep["sampler_class"], rep["model"]) == ("cpu", "TorchNativeAttnBackend", "Sampler", model)
    assert (rep["kv_pool"], rep["allocator"], rep["max_total_num_tokens"]) == ("MHATokenToKVPool", "TokenToKVPoolAllocator", 8192)
    assert len(rep["passes"]) == 6 and all(p["identical"] and p["ref_rms"] > 0.5 for p in rep["passes"]), rep["passes"]
    assert len(rep["sampled"]) == 5 and all(len(s) == 3 for s in rep["sampled"])


@pytest.mark.parametrize("radix", [False, True])
def test_oracle_equals_the_reference_on_the_shared_prefix_job(tmp_path, radix):
    """BASELINE.json's job shape, small, through the reference's ModelRunner (run_shared_prefix_job): group leaders prefilled cold, the
    other requests extending over the leader's slots, `ScheduleBatch.merge_batch`, greedy decode.  radix=True: the prefixes come out
    of the reference's REAL `RadixCache` (`Req.init_next_round_input` -> match_prefix, `cache_unfinished_req` after each prefill,
    `cache_finished_req` at the end) and every non-leader request hits exactly the shared tokens.  Two jobs back to back (the second on
    cleared pools / a reset tree); the oracle reproduces every checked pass bit for bit."""
    _root_or_skip()
    rep = _run("shared-prefix", tmp_path, extra=("--radix",) if radix else ())
    assert len(rep["passes"]) == 15 and all(p["identical"] and p["ref_rms"] > 0.5 for p in rep["passes"]), rep["passes"]
    assert rep["radix_cache"] == ("RadixCache" if radix else None) and rep["radix_hit_lengths"] == ([16] if radix else [])
    assert rep["shape"] == dict(groups=2, per_group=2, prefix=16, unique=8, out=4)


@pytest.mark.parametrize("loop", ["normal", "overlap"])
def test_oracle_equals_the_reference_under_the_references_scheduler(tmp_path, loop):
    """The top of the stack (run_scheduler_job): the reference's `Scheduler` object -- request intake, `PrefillAdder`, the radix cache
    `kv_cache_builder` builds, running-batch merge, `TpModelWorker` / `ModelRunner`, result processing, output streaming -- running
    its OWN `run_event_loop()` -> `event_loop_normal()` / `event_loop_overlap()` (scheduler.py:1696-1853; the server default is the
    overlap loop).  zmq is absent, so one thing is substituted: the receiver's raw socket read hands out the scripted arrivals (group
    leaders first, the other requests once the leaders are decoding) and raises the loop's own `gracefully_exit` when the job is
    done; outputs are taken off `send_to_detokenizer`.  The later requests hit exactly the shared tokens in the radix tree, join the
    running batch (continuous batching), and every request's generated token ids equal the oracle's greedy generation."""
    _root_or_skip()
    rep = _run("scheduler", tmp_path, extra=("--overlap",) if loop == "overlap" else ())
    assert (rep["scheduler"], rep["tp_worker"], rep["tree_cache"], rep["event_loop"]) == ("Scheduler", "TpModelWorker", "UnifiedRadixCache", loop)
    for job in (rep["warm_up"], rep["timed"]):
        b = job["batches_run"]
        assert b["EXTEND x2"] == 2 and b.get("DECODE x4", 0) >= 1 and sum(v for k, v in b.items() if k.startswith("DECODE")) >= 3, b
        assert job["cached_tokens_of_leaders"] == [0] and job["cached_tokens_of_others"] == [16]
        assert job["finished_requests"] == 4 and job["tokens_per_request"] == [4]
    assert rep["oracle"] == dict(requests=4, requests_with_identical_tokens=4, token_agreement=1.0)
    _logits_identical(rep, 16)


def _logits_identical(rep, rows, exact=True):
    """Every forward's `next_token_logits` row, filed by (request, output position) as the scheduler produced it, equals the oracle's
    teacher-forced logits of that position BIT FOR BIT (the CPU run is the reference's torch-native path = the oracle): this pins the
    capture's bookkeeping -- chunks, retraction, overlap steps -- that the GPU twin's band bars rely on."""
    lb = rep["logit_band"]
    assert (lb["rows_compared"], lb["rows_expected"]) == (rows, rows) and lb["logit_rms"] > 0.5, lb
    if exact:
        assert lb["identical_to_the_literal_oracle"], lb
    else:        # a prompt prefilled in chunks attends over its own earlier chunks: another bf16 reduction structure than the oracle's
        assert lb["max_abs_vs_literal"] <= 0.07, lb      # one-shot prefill (in the reference's own backend) -- a couple of logit ulps


def test_oracle_equals_the_reference_under_the_references_scheduler_with_chunked_prefill(tmp_path):
    """The same under `--chunked-prefill-size 64`: the scheduler cuts the 104-token prompts into chunks (EXTEND batches of one, two and
    three requests, each chunk extending over the request's own earlier chunks; the later requests over 80 cached tokens), mixes
    them with the running decode batches, and the token ids still equal the oracle's greedy generation."""
    _root_or_skip()
    rep = _run("scheduler", tmp_path, extra=("--overlap", "--job", "2,3,80,24,6", "--server-args", '{"chunked_prefill_size": 64}'))
    assert rep["chunked_prefill_size"] == 64 and rep["event_loop"] == "overlap"
    for job in (rep["warm_up"], rep["timed"]):
        assert sum(v for k, v in job["batches_run"].items() if k.startswith("EXTEND")) >= 4, job["batches_run"]      # chunks, not two prefills
        assert job["cached_tokens_of_leaders"] == [0] and job["cached_tokens_of_others"] == [80]
        assert job["finished_requests"] == 6 and job["tokens_per_request"] == [6]
    assert rep["oracle"] == dict(requests=6, requests_with_identical_tokens=6, token_agreement=1.0)
    _logits_identical(rep, 36, exact=False)


@pytest.mark.parametrize("extra,what", [(("--logprobs",), "return_logprob + top-2 logprobs on every request"),
                                        (("--dims", "tiny_mixtral"), "MixtralForCausalLM under the scheduler")])
def test_oracle_equals_the_reference_under_the_references_scheduler_logprobs_and_moe(tmp_path, extra, what):
    """(a) every request asks for log-probabilities: the values the scheduler streams with the tokens (sampler ->
    `output_logprob_processor` -> output streamer) equal `log_softmax` of the oracle's logits -- the oracle teacher-forced with the
    produced tokens -- exactly, and so do the top-2 sets; (b) the MoE model under the same loop."""
    _root_or_skip()
    rep = _run("scheduler", tmp_path, extra=("--overlap",) + extra)
    o = rep["oracle"]
    assert (o["requests"], o["requests_with_identical_tokens"], o["token_agreement"]) == (4, 4, 1.0), (what, o)
    if "--logprobs" in extra:
        assert (o["logprob_values"], o["max_abs_logprob_diff"], o["top2_sets_equal"]) == (16, 0.0, 16), o
    _logits_identical(rep, 16)


def test_the_references_scheduler_with_mixed_chunks(tmp_path):
    """`--enable-mixed-chunk --chunked-prefill-size 64`: the scheduler puts a prefill chunk and the running decode requests into ONE
    forward (`ForwardMode.MIXED`).  A decode token then goes through the extend path (one new token over a long prefix), whose bf16
    reduction order differs from the decode path's -- in the reference's own torch-native backend too -- so a near-tie may flip: the
    job's structure is asserted exactly, the tokens against the oracle's decode-path generation for most positions."""
    _root_or_skip()
    rep = _run("scheduler", tmp_path, extra=("--overlap", "--job", "2,3,80,24,6", "--server-args", '{"chunked_prefill_size": 64, "enable_mixed_chunk": true}'))
    for job in (rep["warm_up"], rep["timed"]):
        assert sum(v for k, v in job["batches_run"].items() if k.startswith("MIXED")) >= 2, job["batches_run"]
        assert job["cached_tokens_of_others"] == [80] and job["finished_requests"] == 6 and job["tokens_per_request"] == [6]
    assert rep["oracle"]["token_agreement"] >= 0.8, rep["oracle"]
    if "logit_band" in rep:        # (only when every request produced its tokens as the oracle's teacher forcing expects: 6 x 6 rows)
        assert rep["logit_band"]["rows_compared"] == 36 and rep["logit_band"]["max_abs_vs_literal"] < 0.25, rep["logit_band"]


def test_the_references_scheduler_retracts_and_the_tokens_do_not_change(tmp_path):
    """A KV pool too small for the eight requests' 40 output tokens, admitted aggressively (`--schedule-conservativeness 0.05`): the
    scheduler runs out of slots in the middle of decoding, retracts a request (frees its slots, puts it back into the waiting queue),
    re-prefills it later over whatever the radix tree still holds -- and every request still ends with the oracle's tokens."""
    _root_or_skip()
    rep = _run("scheduler", tmp_path, extra=("--overlap", "--job", "2,4,32,16,40", "--server-args", '{"max_total_tokens": 380, "schedule_conservativeness": 0.05}'))
    assert rep["retracted_requests"] >= 1 and rep["max_total_num_tokens"] == 380
    for job in (rep["warm_up"], rep["timed"]):
        assert job["finished_requests"] == 8 and job["tokens_per_request"] == [40]
    assert rep["oracle"] == dict(requests=8, requests_with_identical_tokens=8, token_agreement=1.0)
    _logits_identical(rep, 320)


def test_the_references_loader_discovers_and_executes_the_plugin(tmp_path):
    import ref_model

    _root_or_skip()
    # from the staged copy when there is one: what the GPU box will import
    env = {"REF_OBJECTS_ROOT": str(ref_model.STAGE)} if (ref_model.STAGE / "sglang").exists() else None
    rep = _run("loader", tmp_path, env)
    from sglang_amd import fused_decode, linear_hook, mem_hooks, position_hooks, tp_hooks
    from sglang_amd.platform import BACKEND_NAME, DISPATCH_KEY

    assert rep["platform"] == "Mi355xSRTPlatform" and rep["out_of_tree"]
    assert rep["dispatch_key"] == DISPATCH_KEY and rep["default_attention_backend"] == BACKEND_NAME
    assert rep["attention_backend_registered"] and rep["attention_backend_choice"]
    assert rep["sampler_registered"] and rep["sampler_choice"]
    assert rep["fused_moe_slot"].endswith("_adapt_fused_func.<locals>.wrapper")
    assert rep["oot_forwards"] == ["DynamicNTKAlphaRotaryEmbedding", "DynamicNTKScalingRotaryEmbedding", "Llama3RotaryEmbedding", "RMSNorm",
                                   "RotaryEmbedding", "SiluAndMul", "TopK", "UnquantizedFusedMoEMethod"]
    targets = sorted(fused_decode.HOOK_TARGETS + tp_hooks.HOOK_TARGETS + position_hooks.HOOK_TARGETS + mem_hooks.HOOK_TARGETS
                     + (linear_hook.HOOK_TARGET, linear_hook.LM_HEAD_HOOK_TARGET))
    assert rep["hooked"] == targets and rep["hooks_applied"] == targets          # every target resolved on the real modules
