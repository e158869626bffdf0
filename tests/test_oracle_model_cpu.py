"""CPU checks of the oracle's own plumbing (oracle/model.py, oracle/parity.py): the batched decode shortcut equals
the reference's per-request loop, the radix-style shared-prefix prefill equals the plain one, and BASELINE.json
configs[0] (Qwen2.5-0.5B greedy decode on the CPU torch-native path) runs end to end."""
import random

import torch

from oracle import ops
from oracle.model import OracleLM
from oracle.parity import LogitStats, bf16_ulp


def _tiny_weights(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    D, H = cfg.head_dim, cfg.hidden_size

    def w(*shape):
        return (torch.randn(shape, generator=g) * 0.05).to(torch.bfloat16)

    out = {"embed_tokens": w(cfg.vocab_size, H), "norm.weight": torch.ones(H, dtype=torch.bfloat16)}
    out["lm_head"] = out["embed_tokens"] if cfg.tie_word_embeddings else w(cfg.vocab_size, H)
    for i in range(cfg.num_hidden_layers):
        p = f"layers.{i}."
        out[p + "input_layernorm.weight"] = torch.ones(H, dtype=torch.bfloat16)
        out[p + "post_attention_layernorm.weight"] = torch.ones(H, dtype=torch.bfloat16)
        nqkv = (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * D
        out[p + "self_attn.qkv_proj.weight"] = w(nqkv, H)
        if cfg.attention_bias:
            out[p + "self_attn.qkv_proj.bias"] = w(nqkv)
        out[p + "self_attn.o_proj.weight"] = w(H, cfg.num_attention_heads * D)
        out[p + "mlp.gate_up_proj.weight"] = w(2 * cfg.intermediate_size, H)
        out[p + "mlp.down_proj.weight"] = w(H, cfg.intermediate_size)
    return out


def test_batched_decode_attention_equals_the_per_request_loop():
    torch.manual_seed(0)
    B, Hq, Hkv, D, kv = 5, 8, 2, 64, 37
    kc = torch.randn(400, Hkv, D).to(torch.bfloat16)
    vc = torch.randn(400, Hkv, D).to(torch.bfloat16)
    r2t = torch.zeros((B + 1, 64), dtype=torch.int32)
    perm = torch.randperm(399)[: B * kv] + 1
    for b in range(B):
        r2t[b + 1, :kv] = perm[b * kv:(b + 1) * kv].to(torch.int32)
    q = torch.randn(B, Hq, D).to(torch.bfloat16)
    pool = torch.arange(1, B + 1)
    lens = torch.full((B,), kv)
    for dt in (None, torch.float32):
        a = ops.decode_attention(q, kc, vc, r2t, pool, lens, D ** -0.5, dt, batched=False)
        b = ops.decode_attention(q, kc, vc, r2t, pool, lens, D ** -0.5, dt, batched=True)
        # fp32: the same arithmetic per request up to the SDPA kernel's blocking; bf16: one output ulp
        torch.testing.assert_close(a.float(), b.float(), atol=2e-6 if dt else 8e-3, rtol=0)
    # ragged lengths must take the loop
    lens2 = lens.clone(); lens2[0] = kv - 3
    a = ops.decode_attention(q, kc, vc, r2t, pool, lens2, D ** -0.5, torch.float32, batched=True)
    b = ops.decode_attention(q, kc, vc, r2t, pool, lens2, D ** -0.5, torch.float32, batched=False)
    assert torch.equal(a, b)


def test_shared_prefix_prefill_of_the_oracle_equals_the_plain_one():
    from sglang_amd.harness.models import CONFIGS

    cfg = CONFIGS["tiny-qwen"]
    w = _tiny_weights(cfg)
    rnd = random.Random(4)
    shared = [rnd.randrange(cfg.vocab_size) for _ in range(20)]
    prompts = [shared + [rnd.randrange(cfg.vocab_size) for _ in range(5)] for _ in range(4)]
    plain = OracleLM(cfg, w, compute_dtype=torch.float32).generate(prompts, 5, return_logits=True)
    reuse = OracleLM(cfg, w, compute_dtype=torch.float32).generate(prompts, 5, return_logits=True, forced=plain[0],
                                                                   share_prefix_groups=[[0, 1], [2, 3]], shared_len=20)
    for a, b in zip(plain[1], reuse[1]):
        # same math; the follower's prefix K/V were computed in the leader's forward (different GEMM row counts)
        torch.testing.assert_close(a, b, atol=2e-2, rtol=0)
    assert reuse[0] == plain[0]


def test_config0_qwen25_05b_greedy_decode_on_the_cpu_torch_native_path():
    """BASELINE.json configs[0]: the whole Qwen2.5-0.5B architecture (24 layers, 14 / 2 heads of 64, qkv bias,
    tied embeddings, vocab 151936) with synthetic weights, prefill + greedy decode through the oracle on CPU."""
    from sglang_amd.harness.models import CONFIGS, synth_weight

    cfg = CONFIGS["qwen2.5-0.5b"]
    H, D = cfg.hidden_size, cfg.head_dim
    w = {"embed_tokens": synth_weight("model.embed_tokens", (cfg.vocab_size, H), "cpu"),
         "norm.weight": torch.ones(H, dtype=torch.bfloat16)}
    w["lm_head"] = w["embed_tokens"]
    nqkv = (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * D
    for i in range(cfg.num_hidden_layers):
        p = f"layers.{i}."
        w[p + "input_layernorm.weight"] = torch.ones(H, dtype=torch.bfloat16)
        w[p + "post_attention_layernorm.weight"] = torch.ones(H, dtype=torch.bfloat16)
        w[p + "self_attn.qkv_proj.weight"] = synth_weight(p + "qkv", (nqkv, H), "cpu")
        w[p + "self_attn.qkv_proj.bias"] = synth_weight(p + "qkv_b", (nqkv,), "cpu")
        w[p + "self_attn.o_proj.weight"] = synth_weight(p + "o", (H, cfg.num_attention_heads * D), "cpu")
        w[p + "mlp.gate_up_proj.weight"] = synth_weight(p + "gu", (2 * cfg.intermediate_size, H), "cpu")
        w[p + "mlp.down_proj.weight"] = synth_weight(p + "down", (H, cfg.intermediate_size), "cpu")
    rnd = random.Random(1)
    prompts = [[rnd.randrange(cfg.vocab_size) for _ in range(12)] for _ in range(2)]
    lm = OracleLM(cfg, w, num_slots=128, max_ctx=64)
    outs, logits = lm.generate(prompts, 4, return_logits=True)
    assert all(len(o) == 4 and all(0 <= t < cfg.vocab_size for t in o) for o in outs)
    assert all(torch.isfinite(l).all() and l.shape == (2, cfg.vocab_size) for l in logits)
    # greedy = arg-max of the recorded logits; a second run reproduces the tokens (deterministic plumbing)
    assert outs == [[int(l[b].argmax()) for l in logits] for b in range(2)]
    assert OracleLM(cfg, w, num_slots=128, max_ctx=64).generate(prompts, 4) == outs


def test_logit_stats_and_bf16_ulp():
    x = torch.tensor([1.0, 1.5, 4.0, 7.9, 0.1])
    assert torch.equal(bf16_ulp(x), torch.tensor([2.0 ** -7, 2.0 ** -7, 2.0 ** -5, 2.0 ** -5, 2.0 ** -11]))
    ref = torch.tensor([[4.0, 1.0, -2.0], [0.5, 0.25, 3.0]])
    got = ref.clone(); got[0, 0] += 2.0 ** -5; got[1, 1] += 1e-4
    st = LogitStats(margin=4.0)
    st.update(0, got, ref)
    s = st.summary()
    assert s["logits_compared"] == 6 and s["frac_bit_identical"] == 4 / 6 and s["frac_within_1e-3"] == 5 / 6
    assert s["frac_within_1_bf16_ulp"] == 1.0 and s["argmax_agreement"] == 1.0 and abs(s["max_abs"] - 2.0 ** -5) < 1e-9
