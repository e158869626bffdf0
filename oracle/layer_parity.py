"""Per-layer / per-stage identical-input parity harness.  TEST INFRASTRUCTURE ONLY (imported by
tests/test_layer_parity_gpu.py and by bench.py's `parity` leg, after its timed region -- never by the product).

See tests/test_layer_parity_gpu.py for what is compared and why the bars are what they are."""
from __future__ import annotations

import random

import torch

from . import ops as oo
from .model import OracleLM, weights_from_product_model
from .parity import bf16_ulp


class _Patch:
    """monkeypatch.setattr for callers outside pytest (bench.py): undo() restores."""

    def __init__(self):
        self._saved = []

    def setattr(self, obj, name, value):
        self._saved.append((obj, name, getattr(obj, name)))
        setattr(obj, name, value)

    def undo(self):
        for obj, name, value in reversed(self._saved):
            setattr(obj, name, value)
        self._saved.clear()


def ulp_stats(got: torch.Tensor, ref: torch.Tensor) -> dict:
    got, ref = got.float(), ref.float()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    g2, r2 = got.reshape(-1, got.shape[-1]), ref.reshape(-1, ref.shape[-1])
    rms = r2.pow(2).mean(-1, keepdim=True).sqrt()
    e = (g2 - r2) / bf16_ulp(torch.maximum(r2.abs(), rms))
    a = e.abs()
    # the same in the plain unit -- the bf16 ulp of the reference value itself, no floor at the row's rms -- reported beside
    # the floored figures so that the floor's effect can be read off (elements that cancel towards zero carry the
    # absolute noise of the sum they came from and count as many "plain" ulps in ANY evaluation, the reference's own too)
    ap = ((g2 - r2) / bf16_ulp(r2)).abs()
    return dict(n=int(a.numel()), frac_identical=float((g2 == r2).float().mean()), frac_within_1ulp=float((a <= 1).float().mean()),
                frac_within_1ulp_plain_unit=float((ap <= 1).float().mean()), frac_within_2ulp_plain_unit=float((ap <= 2).float().mean()),
                frac_within_2ulp=float((a <= 2).float().mean()), max_ulp=float(a.max()), mean_signed_ulp=float(e.mean()),
                rms_ulp=float(e.pow(2).mean().sqrt()), max_abs=float((g2 - r2).abs().max()), ref_rms=float(r2.pow(2).mean().sqrt()))


def _force_topk(layer, ids_getter):
    """Make the product's router return the oracle's expert ids (weights recomputed from the product's own logits)."""
    orig = layer.mlp.topk.forward

    def fwd(hidden_states, router_logits, **kw):
        out = orig(hidden_states, router_logits, **kw)
        ids = ids_getter().to(out.topk_ids.dtype)
        tw = oo.topk_weights_for_ids(router_logits, ids, True).to(out.topk_weights.dtype)
        return type(out)(tw, ids, out.router_logits)

    layer.mlp.topk.forward = fwd


def _fake_fb(extend_lens=None, device=None):
    mode = type("m", (), {"is_extend": staticmethod(lambda: extend_lens is not None)})()
    return type("FB", (), {"forward_mode": mode, "extend_seq_lens": torch.tensor(extend_lens or [0], device=device)})()


def run_layer_parity(cfg, device, prompt_lens, monkeypatch, operator_surface=False, shared_prefix=None):
    """`shared_prefix` = dict(groups, per_group, prefix): the benchmark's geometry -- `groups` x `per_group` requests, the
    first `prefix` tokens of a group's prompts shared, prompt_lens[b] tokens in all.  The product then runs as the bench
    does: a COLD prefill of the group leaders, a WARM prefill of the others over the radix hit (extend over a `prefix`-token
    prefix) and a decode step whose plan finds the groups (shared chunks + private chunks).  The oracle runs the same two
    prefill passes (OracleLM.generate(share_prefix_groups=...): the others extend over the leader's slots, exactly what the
    radix hit means), so every product batch has its own trace record.  Extra phase `prefill_warm` in the result keys;
    `report["_meta"]` says what the radix cache and the decode plan did."""
    from sglang_amd import kernels as K
    from sglang_amd.harness import models as M
    from sglang_amd.harness.engine import Engine, ModelRunner, Req
    from sglang_amd.layers.moe.topk import StandardTopKOutput
    from sglang_amd.layers.rotary_embedding import FusedSetKVBufferArg

    monkeypatch.setattr(M, "OPERATOR_SURFACE_ONLY", operator_surface)
    B = len(prompt_lens)
    rnd = random.Random(7)
    if shared_prefix is None:
        prompts = [[rnd.randrange(cfg.vocab_size) for _ in range(n)] for n in prompt_lens]
    else:
        G, P, PFX = shared_prefix["groups"], shared_prefix["per_group"], shared_prefix["prefix"]
        assert G * P == B and all(n > PFX for n in prompt_lens)
        prompts = []
        for g in range(G):
            sys_p = [rnd.randrange(cfg.vocab_size) for _ in range(PFX)]
            prompts += [sys_p + [rnd.randrange(cfg.vocab_size) for _ in range(prompt_lens[g * P + p] - PFX)] for p in range(P)]
    gen_kw = {}
    if shared_prefix is not None:
        lead = [g * P for g in range(G)]
        rest = [b for b in range(B) if b % P != 0]
        gen_kw = dict(share_prefix_groups=[list(range(g * P, (g + 1) * P)) for g in range(G)], shared_len=PFX)
    total = sum(prompt_lens) + 4 * B + 64
    ctx = max(prompt_lens) + 16
    runner = ModelRunner(cfg, max_total_tokens=total + 1024, max_running_requests=B, max_context_len=ctx, device=device, use_graph=False)
    model = runner.model
    moe = cfg.num_local_experts > 0
    L = cfg.num_hidden_layers
    Hq, Hkv, D = model.num_attention_heads_per_rank, model.num_kv_heads_per_rank, cfg.head_dim
    # ---- the oracle's run: prefill of all prompts + one decode step (the fed token is arbitrary: inputs are replaced)
    weights = weights_from_product_model(model, device)
    oracle = OracleLM(cfg, weights, num_slots=total, max_ctx=ctx, max_reqs=B, device=device, compute_dtype=torch.float32)
    oracle.trace = []
    forced = [[5, 6] for _ in range(B)]
    oracle.generate(prompts, 2, forced=forced, **gen_kw)
    pre, dec = oracle.trace[0], oracle.trace[-1]
    phases = {"prefill": 0} if shared_prefix is None else {"prefill": 0, "prefill_warm": 1}     # trace record of each prefill pass
    assert len(oracle.trace) == len(phases) + 1 and not pre["decode"] and dec["decode"] and len(pre["layers"]) == L
    # ---- the reference against itself: the literal-bf16 oracle from the same per-layer inputs
    literal = OracleLM(cfg, weights, num_slots=total, max_ctx=ctx, max_reqs=B, device=device, compute_dtype=None)
    literal.trace, literal.inject = [], oracle.trace
    if moe:     # the same discrete routing in both evaluations (a near-tie would otherwise move whole rows)
        table = {}
        for i in range(L):
            t = torch.zeros((B, ctx, cfg.num_experts_per_tok), dtype=torch.int32, device=device)
            assert shared_prefix is None
            rows = torch.repeat_interleave(torch.arange(B, device=device), torch.tensor(prompt_lens, device=device))
            t[rows, pre["positions"]] = pre["layers"][i]["topk_ids"]
            t[torch.arange(B, device=device), dec["positions"]] = dec["layers"][i]["topk_ids"]
            table[i] = t
        literal.forced_topk_ids = table
    literal.generate(prompts, 2, forced=forced, **gen_kw)
    noise = {}
    for tag, a, b in [(ph, literal.trace[k], oracle.trace[k]) for ph, k in phases.items()] + [("decode", literal.trace[-1], dec)]:
        bfin = b
        for i in range(L):
            la, lb = a["layers"][i], b["layers"][i]
            noise[f"layer{i}.{tag}.out"] = ulp_stats(la["out"], lb["out"])
            noise[f"layer{i}.{tag}.residual"] = ulp_stats(la["res_out"], lb["res_out"])
            noise[f"layer{i}.{tag}.attn"] = ulp_stats(la["attn_out"], lb["attn_out"])
            # behind the NEXT norm, where the fused decode layer ends
            wn = weights[f"layers.{i + 1}.input_layernorm.weight"] if i + 1 < L else weights["norm.weight"]
            xn, rn = oo.fused_add_rmsnorm(la["out"], la["res_out"], wn, cfg.rms_norm_eps)
            nb = b["layers"][i + 1] if i + 1 < L else None
            noise[f"layer{i}.{tag}.next_normed"] = ulp_stats(xn, nb["normed"] if nb is not None else bfin["final_normed"])
            noise[f"layer{i}.{tag}.next_residual"] = ulp_stats(rn, nb["residual"] if nb is not None else bfin["final_residual"])
    del literal
    report, stages = {}, {}
    state = {"rec": pre, "phase": "prefill"}

    def layer_rec(i):
        return state["rec"]["layers"][i] if i < L else None

    def stage_checks(i, layer, positions, fb, tag):
        """Every operator group of layer i from the oracle's input of that stage."""
        rec = state["rec"]
        lr = layer_rec(i)
        A = layer.self_attn
        pool = fb.token_to_kv_pool
        kb, vb = pool.get_key_buffer(i), pool.get_value_buffer(i)
        loc = fb.out_cache_loc
        T = lr["normed"].shape[0]
        fused = tag == "decode_fused"
        key = f"layer{i}.{tag}"
        bias = A.qkv_proj.bias.data if A.qkv_proj.bias is not None else None
        # 1. qkv_proj -> rope -> KV-row store
        if fused:
            q = K.wstream_qkv_rope(lr["normed"].clone(), A.qkv_proj.weight.data, bias, positions, A.rotary_emb.cos_sin_cache,
                                   Hq, Hkv, D, kb, vb, loc)
        else:
            stages[f"{key}.qkv_proj"] = ulp_stats(A.qkv_proj(lr["normed"].clone()), lr["qkv"])
            q, k, v = (t.contiguous() for t in lr["qkv"].split([Hq * D, Hkv * D, Hkv * D], dim=-1))
            A.rotary_emb(positions, q, k, fused_set_kv_buffer_arg=FusedSetKVBufferArg(value=v, k_buffer=kb, v_buffer=vb, cache_loc=loc))
        stages[f"{key}.rope_q"] = ulp_stats(q.reshape(T, Hq, D), lr["q_rot"].reshape(T, Hq, D))
        stages[f"{key}.stored_k"] = ulp_stats(kb[loc], lr["k_rot"].reshape(T, Hkv, D))
        stages[f"{key}.stored_v"] = ulp_stats(vb[loc], lr["v"].reshape(T, Hkv, D))
        # 2. attention over the oracle's rows (context rows were copied before a decode step)
        kb[loc] = lr["k_rot"].reshape(T, Hkv, D)
        vb[loc] = lr["v"].reshape(T, Hkv, D)
        o = A.attn(lr["q_rot"].contiguous(), None, None, fb, save_kv_cache=False)
        stages[f"{key}.attention"] = ulp_stats(o.reshape(T, Hq, D), lr["attn_out"].reshape(T, Hq, D))
        # 3. o_proj -> residual add + RMSNorm
        attn_in = lr["attn_out"].reshape(T, Hq * D).contiguous()
        if fused:
            res = lr["residual"].clone()
            x = A.o_proj.forward_add_rmsnorm(attn_in, res, layer.post_attention_layernorm)
            stages[f"{key}.o_proj_add_norm"] = ulp_stats(K.unblock(x), lr["post_normed"])
            stages[f"{key}.o_proj_add_norm.residual"] = ulp_stats(res, lr["res_out"])
        else:
            stages[f"{key}.o_proj"] = ulp_stats(A.o_proj(attn_in), lr["o_proj"])
            x, res = layer.post_attention_layernorm(lr["o_proj"].clone(), lr["residual"].clone())
            stages[f"{key}.post_norm"] = ulp_stats(x, lr["post_normed"])
            stages[f"{key}.post_norm.residual"] = ulp_stats(res, lr["res_out"])
        # 4./5. the MLP
        if moe:
            stages[f"{key}.router"] = ulp_stats(layer.mlp.gate(lr["post_normed"].clone()), lr["router_logits"])
            out = layer.mlp.experts(lr["post_normed"].clone(), StandardTopKOutput(lr["topk_weights"].float(), lr["topk_ids"].to(torch.int32),
                                                                                 lr["router_logits"]))
            stages[f"{key}.experts"] = ulp_stats(out, lr["out"])
            return
        act = layer.mlp.gate_up_act(lr["post_normed"].clone(), out_blocked=fused)
        stages[f"{key}.gate_up_silu"] = ulp_stats(K.unblock(act), lr["act"])
        if fused:
            nxt = layer_rec(i + 1)
            norm = model.layers[i + 1].input_layernorm if nxt is not None else model.norm
            res = lr["res_out"].clone()
            x = layer.mlp.down_proj.forward_add_rmsnorm(lr["act"].clone(), res, norm)
            stages[f"{key}.down_add_norm"] = ulp_stats(K.unblock(x), nxt["normed"] if nxt is not None else rec["final_normed"])
            stages[f"{key}.down_add_norm.residual"] = ulp_stats(res, nxt["residual"] if nxt is not None else rec["final_residual"])
        else:
            stages[f"{key}.down_proj"] = ulp_stats(layer.mlp.down_proj(lr["act"].clone()), lr["out"])

    # ---- product layers fed with the oracle's inputs
    def wrap_layer(i, layer):
        orig_fwd, orig_fused = layer.forward, layer.forward_decode_fused

        def fwd(positions, hidden_states, forward_batch, residual):
            rec = state["rec"]
            lr = layer_rec(i)
            tag = "decode_unfused" if rec["decode"] else state["phase"]
            stage_checks(i, layer, positions, forward_batch, tag)
            res_in = lr["res_in"].clone() if lr["res_in"] is not None else None
            h, r = orig_fwd(positions, lr["h_in"].clone(), forward_batch, res_in)
            which = "decode" if rec["decode"] else state["phase"]
            report[f"layer{i}.{which}.out"] = ulp_stats(h, lr["out"])
            report[f"layer{i}.{which}.residual"] = ulp_stats(r, lr["res_out"])
            return h, r

        def fused(positions, normed, forward_batch, residual, next_norm):
            rec = state["rec"]
            lr = layer_rec(i)
            stage_checks(i, layer, positions, forward_batch, "decode_fused")
            res = lr["residual"].clone()
            x = orig_fused(positions, lr["normed"].clone(), forward_batch, res, next_norm)
            nxt = layer_rec(i + 1)
            # the fused layer ends behind the NEXT norm; the unnormed MLP output is compared through the residual
            report[f"layer{i}.decode.next_normed"] = ulp_stats(K.unblock(x), nxt["normed"] if nxt is not None else rec["final_normed"])
            report[f"layer{i}.decode.next_residual"] = ulp_stats(res, nxt["residual"] if nxt is not None else rec["final_residual"])
            residual.copy_(res)
            return x

        layer.forward, layer.forward_decode_fused = fwd, fused
        if moe:
            _force_topk(layer, lambda: state["rec"]["layers"][i]["topk_ids"])

    for i, layer in enumerate(model.layers):
        wrap_layer(i, layer)
    eng = Engine(runner)
    reqs = [Req(b, p, 2) for b, p in enumerate(prompts)]
    r2t_p = runner.req_to_token_pool.req_to_token
    pool = runner.token_to_kv_pool

    def copy_oracle_kv(which):
        """The oracle's K / V rows of requests `which` into the product's slots of the same (request, position)."""
        for b in which:
            n = prompt_lens[b]
            sp = r2t_p[reqs[b].req_pool_idx, :n].long()
            so = oracle.req_to_token[b + 1, :n].long()
            for l in range(L):
                pool.get_key_buffer(l)[sp] = oracle.k_cache[l][so]
                pool.get_value_buffer(l)[sp] = oracle.v_cache[l][so]

    meta = {}
    if shared_prefix is None:
        eng.prefill(reqs)
    else:
        # as bench.py's job(): the leaders first (cold), then the rest over the radix hit (warm)
        eng.prefill([reqs[b] for b in lead])
        copy_oracle_kv(lead)                              # the warm pass attends to the ORACLE's prefix rows
        state["rec"], state["phase"] = oracle.trace[1], "prefill_warm"
        eng.prefill([reqs[b] for b in rest])
        meta["radix_hit_tokens"] = sorted(set(int(reqs[b].cached_tokens) for b in rest))
        # the oracle's members read the leader's prefix slots too (share_prefix_groups): one set of prefix rows per group
        meta["oracle_shares_prefix_slots"] = bool(torch.equal(oracle.req_to_token[1, :PFX], oracle.req_to_token[2, :PFX]))
    # final norm + lm_head from the oracle's last hidden state
    hn, _ = model.norm(pre["layers"][-1]["out"].clone(), pre["layers"][-1]["res_out"].clone())
    stages["final_norm.prefill"] = ulp_stats(hn, pre["final_normed"])
    lg = model.compute_logits(pre["final_normed"].clone(), _fake_fb(prompt_lens if shared_prefix is None else [prompt_lens[b] for b in lead], device))
    stages["lm_head.prefill"] = ulp_stats(lg.next_token_logits, pre["logits"])
    # ---- decode step on the ORACLE's KV rows: copy them to the product's slots of the same (request, position)
    copy_oracle_kv(range(B))
    state["rec"] = dec
    if shared_prefix is not None:                     # the engine's batch order: leaders, then the rest -> back to request order
        eng.running.sort(key=lambda q: q.rid)
        eng._decode_state = None
    assert [q.rid for q in eng.running] == list(range(B))
    eng.decode_step()
    eng.flush_decode_outputs(lag=0)
    if shared_prefix is not None:
        m = getattr(runner.attn_backend, "forward_metadata", None)
        ws = getattr(m, "cascade", None)
        if ws is not None:
            plan = ws.plan.cpu()
            meta["decode_plan"] = dict(groups=int(plan[1]), items=int(plan[0]), shared_kv_tokens=sorted(set(plan[8:8 + B].tolist())))
        meta["decode_contexts"] = [min(prompt_lens) + 1, max(prompt_lens) + 1]
        report["_meta"] = meta
    fused_ran = any(".decode_fused." in k for k in stages)
    # (the fused decode layer takes batches up to 128 rows -- from 65 rows with the wide gate_up projection on the library GEMM;
    # beyond, the operator-by-operator layer.  Sparse-MoE layers have a fused form too since round 5: the attention half + the
    # block's own pieces + add-norm)
    assert fused_ran == (not operator_surface and B <= 128), sorted(stages)
    lg = model.compute_logits(dec["final_normed"].clone(), _fake_fb())
    stages["lm_head.decode"] = ulp_stats(lg.next_token_logits, dec["logits"])
    return report, stages, noise




def summarize(report: dict, stages: dict, noise: dict) -> dict:
    """The figures bench.py's `parity.per_layer` carries."""
    non_attn = {k: v for k, v in stages.items() if not k.endswith(".attention")}
    attn = {k: v for k, v in stages.items() if k.endswith(".attention")}
    return {
        "unit": "bf16 ulps of max(|ref|, row rms); every stage / layer starts from the fp32-accumulating oracle's own inputs",
        "stages": len(stages),
        "stage_min_frac_within_1ulp": min(v["frac_within_1ulp"] for v in stages.values()),
        "stage_max_ulp": max(v["max_ulp"] for v in stages.values()),
        "stage_max_abs_mean_signed_ulp": max(abs(v["mean_signed_ulp"]) for v in stages.values()),
        "gemm_norm_rope_stage_min_frac_bit_identical": min(v["frac_identical"] for v in non_attn.values()),
        "attention_stage_min_frac_within_1ulp": min(v["frac_within_1ulp"] for v in attn.values()),
        "attention_stage_frac_bit_identical": [min(v["frac_identical"] for v in attn.values()), max(v["frac_identical"] for v in attn.values())],
        "whole_layer_rms_ulp": {k: round(v["rms_ulp"], 4) for k, v in report.items()},
        "reference_vs_reference_rms_ulp": {k: round(noise[k]["rms_ulp"], 4) for k in report},
        "whole_layer_max_abs_mean_signed_ulp": max(abs(v["mean_signed_ulp"]) for v in report.values()),
    }
