"""Pin the CPU oracle against (a) the reference's own known-answer tests and
(b) fixtures produced by running the real reference modules
(tests/golden/gen_golden.py).  CPU only."""
import json

import numpy as np
import pytest
import torch

from oracle import host as oh
from oracle import ops as oo


def _load(golden_dir, name):
    return torch.load(golden_dir / name, weights_only=False)


# ---- reference KATs ---------------------------------------------------------
def test_murmur_hash_kat():
    # test/registered/sampling/test_deterministic_gumbel_u1.py:14-21
    h = oh.murmur_hash32(np.array([6469398791980356130], dtype=np.uint64), np.array([7371]), np.array([248146]))
    assert int(h[0, 0]) == 0xFFFFFFFF


def test_gumbel_u1_bucket_does_not_dominate():
    # test_deterministic_gumbel_u1.py:44-51 (u == 1 bucket must not override a ~-52 logprob)
    V, col = 248320, 248146
    logits = torch.zeros(1, V)
    logits[:, col] = -40.0
    probs = torch.softmax(logits, dim=-1)
    tok = oh.sampling_from_probs(probs, torch.tensor([6469398791980356130], dtype=torch.int64), torch.tensor([7371]))
    assert int(tok) != col


def test_kv_indices_reference_scenario():
    # test/registered/attention/test_create_kvindices.py:24-71 shape of the check
    rng = np.random.default_rng(0)
    max_batch, max_ctx = 64, 128
    r2t = np.arange(max_batch * max_ctx, dtype=np.int32).reshape(max_batch, max_ctx)
    for batch in (1, 37):
        pool = rng.choice(max_batch, size=batch, replace=False)
        lens = rng.choice(max_ctx, size=batch, replace=False)
        indptr, idx = oh.create_kv_indices(r2t, pool, lens)
        assert indptr[-1] == lens.sum()
        for b in range(batch):
            np.testing.assert_array_equal(idx[indptr[b]:indptr[b + 1]], r2t[pool[b], :lens[b]])


# ---- fixtures generated from the real reference code ----------------------------
def test_attention_matches_reference(golden_dir):
    cases = _load(golden_dir, "attention_torch_native.pt")
    for name, c in cases.items():
        o = oo.extend_attention(c["q"], c["k_cache"], c["v_cache"], c["req_to_token"], c["req_pool_indices"],
                                c["seq_lens"], c["extend_prefix_lens"], c["extend_seq_lens"], c["scaling"])
        assert torch.equal(o, c["out_extend"]), name
        d = oo.decode_attention(c["q_decode"], c["k_cache"], c["v_cache"], c["req_to_token"], c["req_pool_indices"],
                                c["seq_lens"], c["scaling"])
        assert torch.equal(d, c["out_decode"]), name
        # fp32-accumulating evaluation stays within bf16 rounding of the literal bf16 SDPA
        o32 = oo.extend_attention(c["q"], c["k_cache"], c["v_cache"], c["req_to_token"], c["req_pool_indices"],
                                  c["seq_lens"], c["extend_prefix_lens"], c["extend_seq_lens"], c["scaling"],
                                  compute_dtype=torch.float32)
        torch.testing.assert_close(o32.float(), c["out_extend"].float(), atol=2e-2, rtol=2e-2)
        # sliding-window layers: the restated mask (k <= q and k >= q - W; decode keeps positions [len-1-W, len-1])
        # against the REAL reference's _make_sliding_window_mask + SDPA (torch_native_backend.py:36-48,150-156,251-257).
        # The restatement evaluates the masked form in fp32 and rounds once; the reference ran bf16 SDPA with an
        # attn_mask: equal up to the bf16 rounding of the output.
        W = c["sliding_window"]
        ow = oo.extend_attention(c["q"], c["k_cache"], c["v_cache"], c["req_to_token"], c["req_pool_indices"], c["seq_lens"],
                                 c["extend_prefix_lens"], c["extend_seq_lens"], c["scaling"], sliding_window=W)
        torch.testing.assert_close(ow.float(), c["out_extend_window"].float(), atol=8e-3, rtol=8e-3)
        dw = oo.decode_attention(c["q_decode"], c["k_cache"], c["v_cache"], c["req_to_token"], c["req_pool_indices"],
                                 c["seq_lens"], c["scaling"], sliding_window=W)
        torch.testing.assert_close(dw.float(), c["out_decode_window"].float(), atol=8e-3, rtol=8e-3)
        assert not torch.allclose(ow.float(), c["out_extend"].float(), atol=1e-2)      # the window really bites here


def test_fp8_kv_quantisation_restatement():
    """memory_pool.py:2364-2374: rows hold (K / k_scale) cast to float8_e4m3fn; the attention sees them times the scale."""
    g = torch.Generator().manual_seed(2)
    x = (torch.randn((64, 2, 64), generator=g) * 3).to(torch.bfloat16)
    q = oo.quantize_kv_fp8(x, 0.5)
    assert q.dtype == torch.float8_e4m3fn
    back = q.float() * 0.5
    rel = ((back - x.float()).abs() / x.float().abs().clamp_min(0.05)).max()
    assert float(rel) < 2.0 ** -3                      # 3 significand bits: half an ulp = 2^-4 relative, plus the bf16 divide
    # literal reference arithmetic: cache_k.div_(k_scale) on the bf16 tensor, then .to(dtype)
    lit = x.clone(); lit.div_(0.5)
    assert torch.equal(lit.to(torch.float8_e4m3fn).view(torch.uint8), q.view(torch.uint8))
    # saturation instead of NaN beyond the e4m3 range
    big = torch.tensor([1000.0, -1000.0, 448.0], dtype=torch.bfloat16)
    assert oo.quantize_kv_fp8(big).float().tolist() == [448.0, -448.0, 448.0]


def test_norm_rope_match_reference(golden_dir):
    g = _load(golden_dir, "elementwise_native.pt")
    for name, c in g.items():
        if name.startswith("rmsnorm"):
            assert torch.equal(oo.rmsnorm(c["x"], c["weight"], c["eps"]), c["out"]), name
            y, r = oo.fused_add_rmsnorm(c["x"], c["residual"], c["weight"], c["eps"])
            assert torch.equal(y, c["out_fused"]) and torch.equal(r, c["residual_out"]), name
        else:
            if c["llama3"]:
                inv = oo.llama3_inv_freq(c["head_size"], c["base"], 8.0, 1.0, 4.0, 8192)
            else:
                inv = oo.rope_inv_freq(c["head_size"], c["base"])
            assert torch.equal(inv, c["inv_freq"]), name
            cache = oo.cos_sin_cache(inv, c["max_pos"])
            assert torch.equal(cache, c["cache_f32"]), name
            q, k = oo.rotary_embedding(c["positions"], c["q"], c["k"], c["head_size"], c["cache"], c["is_neox"])
            assert torch.equal(q, c["q_out"]) and torch.equal(k, c["k_out"]), name


def test_moe_matches_reference(golden_dir):
    c = _load(golden_dir, "moe_native.pt")
    tw, ti = oo.fused_topk(c["router_logits"], c["topk"], True)
    assert torch.equal(tw, c["topk_weights"]) and torch.equal(ti.long(), c["topk_ids"].long())
    y = oo.moe_forward(c["x"], c["w13"], c["w2"], tw, ti)
    # the reference holds two native forms (einsum / per-expert loop) that differ by bf16 rounding
    torch.testing.assert_close(y.float(), c["out_einsum"].float(), atol=3e-2, rtol=3e-2)


def test_sampler_kept_set_matches_reference(golden_dir):
    c = _load(golden_dir, "sampler_torch.pt")
    B = c["probs"].shape[0]
    for need_min_p in (False, True):
        seeds = None if need_min_p else torch.arange(B, dtype=torch.int64) + 1
        _, kept, idx = oh.top_k_top_p_min_p_sampling_from_probs(
            c["probs"].clone(), c["top_ks"], c["top_ps"], c["min_ps"], need_min_p, seeds,
            torch.zeros(B, dtype=torch.int64), return_kept=True) if seeds is not None else (None,) * 3
        if kept is None:
            torch.manual_seed(0)
            _, kept, idx = oh.top_k_top_p_min_p_sampling_from_probs(
                c["probs"].clone(), c["top_ks"], c["top_ps"], c["min_ps"], True, None,
                torch.zeros(B, dtype=torch.int64), return_kept=True)
        assert torch.equal(kept, c[f"kept_sorted_minp{int(need_min_p)}"])
        assert torch.equal(idx[:, 0], c[f"rank0_ids_minp{int(need_min_p)}"])


def test_host_int_matches_reference(golden_dir):
    g = json.loads((golden_dir / "host_int.json").read_text())
    for c in g["alloc_extend"]:
        out, used = oh.alloc_extend(c["prefix"], c["seq"], c["last_loc"], c["free_pages"], c["page_size"])
        assert out.tolist() == c["out"], c
    for c in g["compute_position"]:
        p, s = oh.compute_position(c["prefix"], c["extend"])
        assert p.tolist() == c["positions"] and s.tolist() == c["start"]
    r2t = np.arange(5 * 11, dtype=np.int32).reshape(5, 11)
    c = g["get_last_loc"]
    assert oh.get_last_loc(r2t, c["req_pool"], c["prefix"]).tolist() == c["out"]
    c = g["clamp_position"]
    assert oh.clamp_position(c["seq"]).tolist() == c["out"]


def test_bruteforce_prefix_cache_matches_reference_trace(golden_dir):
    traces = json.loads((golden_dir / "radix_trace.json").read_text())
    assert traces["main_scenario"]["match"] == [1, 2, 3]
    # match/insert results must agree with the brute-force model until the first eviction
    for page in (1, 4):
        model = oh.BruteForcePrefixCache(page_size=page)
        for op in traces[f"page{page}"]:
            if op["op"] == "evict":
                break
            if op["op"] == "insert":
                assert model.insert(op["ids"], op["vals"]) == op["prefix_len"]
            elif op["op"] == "match":
                assert model.match(op["ids"]) == op["indices"]


# --------------------------------------------------------------------------------------------------------------
# Logit soft cap, custom (verify) mask and sliding window pinned against the REFERENCE'S OWN Triton kernels
# (tests/golden/attention_triton.pt: outputs of extend_attention_fwd / decode_attention_fwd run under triton-rocm on an
# MI355X by tests/golden/gen_triton_golden.py).  The Triton kernels round P to bf16 before the PV product and write
# bf16, the oracle evaluates in fp32 and rounds once: the two agree to one bf16 ulp of the output.
def _triton_cases(golden_dir):
    cases = torch.load(golden_dir / "attention_triton.pt")
    return {k: v for k, v in cases.items() if not k.startswith("_")}


def _close_1ulp(got, ref, what, max_abs=2.0 ** -7, rms=8e-4):
    e = (got.float() - ref.float()).abs()
    assert float(e.max()) <= max_abs * max(1.0, float(ref.float().abs().max())), (what, float(e.max()))
    assert float(e.pow(2).mean().sqrt()) <= rms, (what, float(e.pow(2).mean().sqrt()))


def test_oracle_extend_cap_window_match_reference_triton(golden_dir):
    for name, d in _triton_cases(golden_dir).items():
        kw = dict(k_cache=d["k_cache"], v_cache=d["v_cache"], req_to_token=d["req_to_token"], req_pool_indices=d["req_pool_indices"],
                  seq_lens=d["seq_lens"], extend_prefix_lens=d["extend_prefix_lens"], extend_seq_lens=d["extend_seq_lens"],
                  scaling=d["scaling"], compute_dtype=torch.float32)
        cap, win = d["logit_cap"], d["sliding_window"]
        for tag, opt in dict(causal={}, cap=dict(logit_cap=cap), window=dict(sliding_window=win),
                             cap_window=dict(logit_cap=cap, sliding_window=win)).items():
            _close_1ulp(oo.extend_attention(d["q"], **kw, **opt), d["out_extend_" + tag], f"{name} extend {tag}")
        # the cap must matter on these inputs, or the case would pin nothing
        assert float((d["out_extend_cap"].float() - d["out_extend_causal"].float()).abs().max()) > 0.05


def test_oracle_verify_mask_matches_reference_triton(golden_dir):
    for name, d in _triton_cases(golden_dir).items():
        v = d["verify"]
        kw = dict(k_cache=v["k_cache"], v_cache=v["v_cache"], req_to_token=v["req_to_token"], req_pool_indices=v["req_pool_indices"],
                  seq_lens=v["seq_lens"], extend_prefix_lens=v["extend_prefix_lens"], extend_seq_lens=v["extend_seq_lens"],
                  scaling=d["scaling"], compute_dtype=torch.float32)
        for tag in ("verify", "verify_prefix_masked"):
            for cap in (0.0, d["logit_cap"]):
                o = oo.extend_attention(v["q"], custom_mask=d[tag + "_mask"], mask_indptr=d[tag + "_mask_indptr"].tolist(), logit_cap=cap, **kw)
                _close_1ulp(o, d["out_" + tag + ("_cap" if cap else "")], f"{name} {tag} cap={cap}")
        # a tree mask is not the causal mask: the fixture must tell them apart
        o_causal = oo.extend_attention(v["q"], **kw)
        assert float((o_causal.float() - d["out_verify"].float()).abs().max()) > 0.05


def test_oracle_decode_cap_matches_reference_triton(golden_dir):
    for name, d in _triton_cases(golden_dir).items():
        kw = dict(k_cache=d["k_cache"], v_cache=d["v_cache"], req_to_token=d["req_to_token"], req_pool_indices=d["req_pool_indices"],
                  seq_lens=d["seq_lens"], scaling=d["scaling"], compute_dtype=torch.float32)
        # the reference's MHA decode kernel (kv_group_num == 1, decode_attention.py:362 `tl.sum(q[None, :] * k, 1)`)
        # multiplies and sums the scores in bf16: at |score| ~ 40 that is a score error of ~0.1 -- its own outputs sit
        # 6e-2 from an fp32 evaluation; the grouped kernel (tl.dot, fp32) pins the cap to one output ulp
        loose = d["q_decode"].shape[1] == d["k_cache"].shape[1]
        bars = dict(max_abs=0.08, rms=1e-2) if loose else {}
        _close_1ulp(oo.decode_attention(d["q_decode"], **kw), d["out_decode"], f"{name} decode", **bars)
        _close_1ulp(oo.decode_attention(d["q_decode"], logit_cap=d["logit_cap"], **kw), d["out_decode_cap"], f"{name} decode cap", **bars)
        assert float((d["out_decode_cap"].float() - d["out_decode"].float()).abs().max()) > 0.05
