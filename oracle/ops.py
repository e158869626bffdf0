"""Floating-point operator oracles (plain torch ops), restating the reference's
`forward_native` / torch-native implementations.  TEST INFRASTRUCTURE ONLY.

Device-agnostic: on CPU tensors this is the reference's CPU torch-native path; on HIP
tensors it is what the reference's `torch_native` attention backend + forward_native
operators execute on a GPU (torch SDPA / F.linear / elementwise ops, no custom kernel),
which is how the full-size configurations are checked in seconds.

All paths are relative to /root/reference/python/sglang.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------- RMSNorm
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """kernels/ops/layernorm/__init__.py:75-92 (RMSNormOp.forward_native)."""
    xf = x.to(torch.float32)
    variance = xf.pow(2).mean(dim=-1, keepdim=True)
    xf = xf * torch.rsqrt(variance + eps)
    return (xf * weight).to(x.dtype)


def fused_add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float
                      ) -> Tuple[torch.Tensor, torch.Tensor]:
    """srt/layers/layernorm.py:777-826 (RMSNorm.forward_native with residual):
    x = x.float() + residual.float(); residual = x.to(dtype); norm on fp32 x;
    (x * weight).to(dtype)."""
    orig = x.dtype
    xf = x.to(torch.float32) + residual.to(torch.float32)
    new_residual = xf.to(orig)
    variance = xf.pow(2).mean(dim=-1, keepdim=True)
    xf = xf * torch.rsqrt(variance + eps)
    return (xf * weight).to(orig), new_residual


# ---------------------------------------------------------------- activation
def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    """srt/layers/activation.py:141-143 (SiluAndMul.forward_native)."""
    d = x.shape[-1] // 2
    return F.silu(x[..., :d]) * x[..., d:]


# ---------------------------------------------------------------- RoPE
def rope_inv_freq(rotary_dim: int, base: float) -> torch.Tensor:
    """srt/layers/rotary_embedding/base.py:153-171 (_compute_inv_freq)."""
    return 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))


def llama3_inv_freq(rotary_dim: int, base: float, scaling_factor: float, low_freq_factor: float,
                    high_freq_factor: float, orig_max_position: int) -> torch.Tensor:
    """srt/layers/rotary_embedding/rope_variant.py:560-580 (Llama3RotaryEmbedding)."""
    inv_freqs = rope_inv_freq(rotary_dim, base)
    low_freq_wavelen = orig_max_position / low_freq_factor
    high_freq_wavelen = orig_max_position / high_freq_factor
    wave_len = 2 * math.pi / inv_freqs
    if low_freq_factor != high_freq_factor:
        smooth = (orig_max_position / wave_len - low_freq_factor) / (high_freq_factor - low_freq_factor)
    else:
        smooth = 0
    return torch.where(
        wave_len < high_freq_wavelen,
        inv_freqs,
        torch.where(wave_len > low_freq_wavelen, inv_freqs / scaling_factor,
                    (1 - smooth) * inv_freqs / scaling_factor + smooth * inv_freqs),
    )


def cos_sin_cache(inv_freq: torch.Tensor, max_position: int) -> torch.Tensor:
    """base.py:173-182 (_compute_cos_sin_cache): fp32 [max_pos, rot] = cos || sin."""
    t = torch.arange(max_position, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1)


def _apply_rotary_emb(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, is_neox_style: bool) -> torch.Tensor:
    """srt/layers/rotary_embedding/utils.py:36-63: cos/sin are cast to x.dtype first."""
    cos = cos.unsqueeze(-2).to(x.dtype)
    sin = sin.unsqueeze(-2).to(x.dtype)
    if is_neox_style:
        x1, x2 = torch.chunk(x, 2, dim=-1)
    else:
        x1 = x[..., ::2]
        x2 = x[..., 1::2]
    o1 = x1 * cos - x2 * sin
    o2 = x2 * cos + x1 * sin
    if is_neox_style:
        return torch.cat((o1, o2), dim=-1)
    return torch.stack((o1, o2), dim=-1).flatten(-2)


def rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, head_size: int,
                     cache: torch.Tensor, is_neox_style: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """base.py:236-276 (RotaryEmbedding.forward_native); rotary_dim = cache.shape[-1]."""
    rotary_dim = cache.shape[-1]
    positions = positions.flatten()
    num_tokens = positions.shape[0]
    cos_sin = cache.index_select(0, positions)
    cos, sin = cos_sin.chunk(2, dim=-1)

    def one(t: torch.Tensor) -> torch.Tensor:
        shape = t.shape
        t = t.reshape(num_tokens, -1, head_size)
        rot = _apply_rotary_emb(t[..., :rotary_dim], cos, sin, is_neox_style)
        return torch.cat((rot, t[..., rotary_dim:]), dim=-1).reshape(shape)

    return one(query), one(key)


# ---------------------------------------------------------------- KV store
def store_kv(k: torch.Tensor, v: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, loc: torch.Tensor) -> None:
    """srt/mem_cache/memory_pool.py:189-193 (naive path: k_cache[indices] = k)."""
    k_cache[loc] = k.view(k.shape[0], *k_cache.shape[1:])
    v_cache[loc] = v.view(v.shape[0], *v_cache.shape[1:])


# ---------------------------------------------------------------- KV formats
FP8_MAX = 448.0


def quantize_kv_fp8(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """srt/mem_cache/memory_pool.py:2364-2374: `cache_k.div_(k_scale)` on the bf16 rows, then the cast to the
    pool dtype (float8_e4m3fn, OCP).  Out-of-range values are clamped to +-448 first (torch's cast would produce
    NaN there; the product saturates -- the only place the two could differ is excluded from the comparison)."""
    y = x.clone()
    if scale != 1.0:
        y.div_(scale)
    return y.float().clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)


def _kv_rows(cache: torch.Tensor, toks: torch.Tensor, descale: float, like: torch.Tensor) -> torch.Tensor:
    """Rows `toks` of a [slots, H, D] pool as compute values: fp8 rows are widened and multiplied by the scale
    (triton_backend.py:1418-1420 k_descale / v_descale), bf16 rows pass through."""
    rows = cache[toks]
    if rows.dtype == torch.float8_e4m3fn:
        return rows.to(torch.float32) * descale
    return rows


def _manual_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scaling: float, mask: Optional[torch.Tensor],
                      logit_cap: float) -> torch.Tensor:
    """softmax(cap * tanh(q k^T * scaling / cap) masked) v in fp32 (extend_attention.py:546-547 for the cap, which
    SDPA cannot express).  q [Hq, Tq, D], k / v [Hkv, Tk, D], mask [Tq, Tk] bool (True = attend)."""
    Hq, Hkv = q.shape[0], k.shape[0]
    qf, kf, vf = q.float(), k.float(), v.float()
    if Hq != Hkv:
        kf = kf.repeat_interleave(Hq // Hkv, dim=0)
        vf = vf.repeat_interleave(Hq // Hkv, dim=0)
    s = torch.matmul(qf, kf.transpose(1, 2)) * scaling
    if logit_cap > 0:
        s = logit_cap * torch.tanh(s / logit_cap)
    if mask is not None:
        s = s.masked_fill(~mask.unsqueeze(0), float("-inf"))
    pr = torch.softmax(s, dim=-1)
    pr = torch.nan_to_num(pr, nan=0.0)          # a fully masked row attends to nothing
    return torch.matmul(pr, vf)


# ---------------------------------------------------------------- attention
def extend_attention(query: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, req_to_token: torch.Tensor,
                     req_pool_indices: torch.Tensor, seq_lens: torch.Tensor, extend_prefix_lens: torch.Tensor,
                     extend_seq_lens: torch.Tensor, scaling: float, causal: bool = True,
                     compute_dtype: Optional[torch.dtype] = None, k_scale: float = 1.0, v_scale: float = 1.0,
                     sliding_window: int = -1, logit_cap: float = 0.0, custom_mask: Optional[torch.Tensor] = None,
                     mask_indptr: Optional[Sequence[int]] = None) -> torch.Tensor:
    """srt/layers/attention/torch_native_backend.py:61-174 (_run_sdpa_forward_extend).

    query [T, Hq, D]; caches [slots, Hkv, D].  The reference pads Q to the kv
    length, runs SDPA with is_causal and keeps rows [prefix:].  compute_dtype
    = torch.float32 evaluates the same graph in fp32 on the bf16-rounded
    inputs (the fp32-accumulation reference of SURVEY section 8(c))."""
    out = torch.empty_like(query)
    q = query.movedim(0, query.dim() - 2)  # [H, T, D]
    enable_gqa = query.shape[1] != k_cache.shape[1]
    start_q = 0
    ext_l, pre_l, kv_l, pool_l = (t.tolist() for t in (extend_seq_lens, extend_prefix_lens, seq_lens, req_pool_indices))
    for i in range(seq_lens.shape[0]):
        ext = int(ext_l[i])
        pre = int(pre_l[i])
        kv = int(kv_l[i])
        end_q = start_q + ext
        per_req_query = q[:, start_q:end_q, :]
        red = torch.empty((per_req_query.shape[0], kv, per_req_query.shape[2]), dtype=per_req_query.dtype,
                          device=query.device)
        red.zero_()  # the reference leaves the padded rows uninitialised; they are discarded
        red[:, pre:, :] = per_req_query
        toks = req_to_token[int(pool_l[i]), :kv].long()
        key = _kv_rows(k_cache, toks, k_scale, query).movedim(0, query.dim() - 2)
        val = _kv_rows(v_cache, toks, v_scale, query).movedim(0, query.dim() - 2)
        special = sliding_window >= 0 or logit_cap > 0 or custom_mask is not None
        if special:
            # rows [pre, kv) of the padded query are the real ones (torch_native_backend.py:61-174); masks are over
            # absolute positions: causal k <= q, window k >= q - W (:36-48), or the verify mask (extend_attention.py
            # USE_CUSTOM_MASK: it replaces the causal rule, the prefix part is AND-ed with "inside the prefix")
            q_pos = torch.arange(pre, kv, device=query.device).unsqueeze(1)
            k_pos = torch.arange(kv, device=query.device).unsqueeze(0)
            if custom_mask is not None:
                m0 = int(mask_indptr[i])
                mask = custom_mask[m0: m0 + ext * kv].view(ext, kv).to(torch.bool)
            else:
                mask = (k_pos <= q_pos) if causal else torch.ones((ext, kv), dtype=torch.bool, device=query.device)
            if sliding_window >= 0:
                mask = mask & (k_pos >= q_pos - sliding_window)
            o = _manual_attention(per_req_query, key, val, scaling, mask, logit_cap)      # [H, ext, D]
            out[start_q:end_q] = o.movedim(query.dim() - 2, 0).to(out.dtype)
            start_q = end_q
            continue
        if key.dtype != red.dtype and compute_dtype is None:
            key, val = key.to(red.dtype), val.to(red.dtype)
        if compute_dtype is not None:
            red, key, val = red.to(compute_dtype), key.to(compute_dtype), val.to(compute_dtype)
        o = F.scaled_dot_product_attention(red.unsqueeze(0), key.unsqueeze(0), val.unsqueeze(0),
                                           enable_gqa=enable_gqa, scale=scaling, is_causal=causal)
        o = o.squeeze(0).movedim(query.dim() - 2, 0)
        out[start_q:end_q] = o[pre:].to(out.dtype)
        start_q = end_q
    return out


def decode_attention(query: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, req_to_token: torch.Tensor,
                     req_pool_indices: torch.Tensor, seq_lens: torch.Tensor, scaling: float,
                     compute_dtype: Optional[torch.dtype] = None, batched: bool = False, k_scale: float = 1.0,
                     v_scale: float = 1.0, sliding_window: int = -1, logit_cap: float = 0.0) -> torch.Tensor:
    """torch_native_backend.py:176-277 (_run_sdpa_forward_decode): one query per request."""
    out = torch.empty_like(query)
    q = query.movedim(0, query.dim() - 2)
    enable_gqa = query.shape[1] != k_cache.shape[1]
    kv_l, pool_l = seq_lens.tolist(), req_pool_indices.tolist()
    special = sliding_window >= 0 or logit_cap > 0 or k_cache.dtype == torch.float8_e4m3fn
    if batched and len(set(kv_l)) == 1 and len(kv_l) > 1 and not special:
        # every request has the same KV length: the per-request SDPA calls of the reference loop are issued as one
        # batched call (identical arithmetic per request; tests/test_oracle_golden.py checks it against the loop)
        kv = int(kv_l[0])
        toks = req_to_token[req_pool_indices.long(), :kv].long()          # [B, kv]
        key = k_cache[toks].transpose(1, 2)                               # [B, Hkv, kv, D]
        val = v_cache[toks].transpose(1, 2)
        qb = query.unsqueeze(2)                                           # [B, Hq, 1, D]
        if compute_dtype is not None:
            qb, key, val = qb.to(compute_dtype), key.to(compute_dtype), val.to(compute_dtype)
        o = F.scaled_dot_product_attention(qb, key, val, enable_gqa=enable_gqa, scale=scaling, is_causal=False)
        return o.squeeze(2).to(out.dtype)
    for i in range(seq_lens.shape[0]):
        kv = int(kv_l[i])
        per_req_query = q[:, i:i + 1, :]
        # the query sits at position kv - 1; a window keeps kv positions [kv - 1 - W, kv - 1] (torch_native_backend.py:36-48)
        k0 = max(0, kv - 1 - sliding_window) if sliding_window >= 0 else 0
        toks = req_to_token[int(pool_l[i]), k0:kv].long()
        key = _kv_rows(k_cache, toks, k_scale, query).movedim(0, query.dim() - 2)
        val = _kv_rows(v_cache, toks, v_scale, query).movedim(0, query.dim() - 2)
        if logit_cap > 0:
            out[i:i + 1] = _manual_attention(per_req_query, key, val, scaling, None, logit_cap).movedim(query.dim() - 2, 0).to(out.dtype)
            continue
        if key.dtype != per_req_query.dtype and compute_dtype is None:
            key, val = key.to(per_req_query.dtype), val.to(per_req_query.dtype)
        if compute_dtype is not None:
            per_req_query, key, val = per_req_query.to(compute_dtype), key.to(compute_dtype), val.to(compute_dtype)
        o = F.scaled_dot_product_attention(per_req_query.unsqueeze(0), key.unsqueeze(0), val.unsqueeze(0),
                                           enable_gqa=enable_gqa, scale=scaling, is_causal=False)
        out[i:i + 1] = o.squeeze(0).movedim(query.dim() - 2, 0).to(out.dtype)
    return out


# ---------------------------------------------------------------- MoE
_RENORMALIZE_SUM_EPSILON = 1e-20  # srt/layers/moe/topk.py (module constant)


def fused_topk(gating_output: torch.Tensor, topk: int, renormalize: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """srt/layers/moe/topk.py:690-736 (fused_topk_torch_native, softmax scoring, no bias)."""
    w = gating_output.float().softmax(dim=-1)
    w, ids = torch.topk(w, topk, dim=-1)
    if renormalize:
        w = w / (w.sum(dim=-1, keepdim=True, dtype=torch.float32) + _RENORMALIZE_SUM_EPSILON)
    return w, ids.to(torch.int32)


def topk_weights_for_ids(gating_output: torch.Tensor, ids: torch.Tensor, renormalize: bool) -> torch.Tensor:
    """fused_topk's weights (topk.py:690-736) for a GIVEN expert choice: softmax scores gathered at `ids`, renormalised."""
    w = gating_output.float().softmax(dim=-1).gather(1, ids.to(torch.int64))
    if renormalize:
        w = w / (w.sum(dim=-1, keepdim=True, dtype=torch.float32) + _RENORMALIZE_SUM_EPSILON)
    return w


def moe_forward(x: torch.Tensor, w13: torch.Tensor, w2: torch.Tensor, topk_weights: torch.Tensor,
                topk_ids: torch.Tensor) -> torch.Tensor:
    """srt/layers/moe/fused_moe_native.py:61-164 (moe_forward_native, silu, no bias):
    per-expert F.linear -> SiluAndMul -> F.linear, combine in topk_weights.dtype."""
    E = w13.shape[0]
    cnts = topk_ids.new_zeros((topk_ids.shape[0], E))
    cnts.scatter_(1, topk_ids.to(torch.int64), 1)
    tokens_per_expert = cnts.sum(dim=0)
    idxs = topk_ids.view(-1).argsort()
    sorted_tokens = x[idxs // topk_ids.shape[1]]
    outputs = []
    start = 0
    for i, n in enumerate(tokens_per_expert.tolist()):
        if n == 0:
            continue
        t = sorted_tokens[start:start + n]
        gate_up = F.linear(t, w13[i])
        act = silu_and_mul(gate_up)
        outputs.append(F.linear(act, w2[i]))
        start += n
    outs = torch.cat(outputs, dim=0) if outputs else sorted_tokens.new_empty(0)
    new_x = torch.empty_like(outs)
    new_x[idxs] = outs
    return (new_x.view(*topk_ids.shape, -1).type(topk_weights.dtype)
            .mul_(topk_weights.unsqueeze(dim=-1)).sum(dim=1).type(new_x.dtype))
