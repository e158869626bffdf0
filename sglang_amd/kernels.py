"""Thin torch-tensor wrappers over the C ABI (pointer + stride plumbing only).

Every function enqueues on torch's current HIP stream, so the kernels compose
with torch ops and are captured by torch.cuda.graph (hipGraph) like any other
launch.  No function here computes anything on the host or falls back to torch.
"""
from __future__ import annotations

import functools
from typing import Optional, Tuple

import torch

from . import native

_BF16 = torch.bfloat16


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(cond: bool, msg: str) -> None:
    if not cond:
        raise ValueError(msg)


def _kv_rows_dtype_ok(cache: torch.Tensor, kv_fp8: bool) -> bool:
    """bf16 rows, or OCP e4m3 rows held either as raw bytes (this package's pool: `store_dtype` uint8) or as the float8_e4m3fn
    VIEW of them that the reference's pool hands out (memory_pool.py:2288-2290 `k_buffer[...].view(self.dtype)`): same bytes."""
    if kv_fp8:
        return cache.dtype in (torch.uint8, torch.float8_e4m3fn)
    return cache.dtype == _BF16


def _dev(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "sglang_amd kernels need tensors on a HIP device (got a CPU tensor); "
                "there is no CPU fallback"
            )


# --------------------------------------------------------------------------- norm
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sgl_kernel.rmsnorm(input, weight, eps, out) equivalent."""
    _dev(x, weight)
    _need(x.dtype == _BF16 and weight.dtype == _BF16, "rmsnorm: bf16 only")
    hidden = x.shape[-1]
    x2 = x.reshape(-1, hidden)
    _need(x2.stride(-1) == 1, "rmsnorm: last dim must be contiguous")
    if out is None:
        out = torch.empty_like(x2)
    o2 = out.reshape(-1, hidden)
    native.call("sgl_amd_rmsnorm", x2.data_ptr(), weight.data_ptr(), o2.data_ptr(), x2.shape[0], hidden,
                x2.stride(0), o2.stride(0), float(eps), _stream())
    return out.view(x.shape) if out.shape != x.shape else out


def embedding_rmsnorm(ids: torch.Tensor, table: torch.Tensor, weight: torch.Tensor, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(table[ids], RMSNorm(table[ids]) * weight): the decode step's embedding lookup and the first layer's input norm in one
    launch (bf16 table [V, H], int64 ids [M])."""
    _dev(ids, table, weight)
    _need(ids.dtype == torch.int64 and ids.dim() == 1 and ids.is_contiguous(), "embedding_rmsnorm: int64 ids [M]")
    _need(table.dtype == _BF16 and table.dim() == 2 and table.stride(1) == 1 and weight.dtype == _BF16 and weight.numel() == table.shape[1],
          "embedding_rmsnorm: bf16 table [V, H] and weight [H]")
    M, H = ids.shape[0], table.shape[1]
    hidden = torch.empty((M, H), dtype=_BF16, device=table.device)
    out = torch.empty((M, H), dtype=_BF16, device=table.device)
    native.call("sgl_amd_embedding_rmsnorm", ids.data_ptr(), table.data_ptr(), weight.data_ptr(), hidden.data_ptr(), out.data_ptr(),
                M, H, table.shape[0], table.stride(0), hidden.stride(0), out.stride(0), float(eps), _stream())
    return hidden, out


def fused_add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float) -> None:
    """sgl_kernel.fused_add_rmsnorm(input, residual, weight, eps): both updated in place."""
    _dev(x, residual, weight)
    _need(x.dtype == _BF16 and residual.dtype == _BF16 and weight.dtype == _BF16, "fused_add_rmsnorm: bf16 only")
    hidden = x.shape[-1]
    x2 = x.view(-1, hidden)
    r2 = residual.view(-1, hidden)
    _need(x2.stride(-1) == 1 and r2.stride(-1) == 1, "fused_add_rmsnorm: last dim must be contiguous")
    native.call("sgl_amd_fused_add_rmsnorm", x2.data_ptr(), r2.data_ptr(), weight.data_ptr(), x2.shape[0], hidden,
                x2.stride(0), r2.stride(0), float(eps), _stream())


# --------------------------------------------------------------------- activation
def silu_and_mul(x: torch.Tensor, out: Optional[torch.Tensor] = None, round_intermediate: bool = True) -> torch.Tensor:
    _dev(x)
    _need(x.dtype == _BF16, "silu_and_mul: bf16 only")
    d = x.shape[-1] // 2
    x2 = x.reshape(-1, 2 * d)
    _need(x2.stride(-1) == 1, "silu_and_mul: last dim must be contiguous")
    if out is None:
        out = torch.empty(x.shape[:-1] + (d,), dtype=x.dtype, device=x.device)
    o2 = out.view(-1, d)
    native.call("sgl_amd_silu_and_mul", x2.data_ptr(), o2.data_ptr(), x2.shape[0], d, x2.stride(0), o2.stride(0),
                1 if round_intermediate else 0, _stream())
    return out


# --------------------------------------------------------------------------- rope
def rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, head_size: int,
                     cos_sin_cache: torch.Tensor, is_neox: bool = True,
                     value: Optional[torch.Tensor] = None, k_cache: Optional[torch.Tensor] = None,
                     v_cache: Optional[torch.Tensor] = None, cache_loc: Optional[torch.Tensor] = None) -> None:
    """In-place rope on query/key ([T, H*D] or [T, H, D]); optional fused KV store."""
    _dev(positions, query, key, cos_sin_cache)
    _need(query.dtype == _BF16 and key.dtype == _BF16, "rotary_embedding: bf16 only")
    _need(positions.dtype == torch.int64, "rotary_embedding: positions must be int64")
    _need(cos_sin_cache.dtype in (_BF16, torch.float32) and cos_sin_cache.is_contiguous(), "rotary_embedding: cache dtype")
    T = positions.numel()
    q2 = query.view(T, -1)
    k2 = key.view(T, -1)
    _need(q2.stride(-1) == 1 and k2.stride(-1) == 1, "rotary_embedding: last dim must be contiguous")
    hq = q2.shape[1] // head_size
    hk = k2.shape[1] // head_size
    rot = cos_sin_cache.shape[-1]
    fused = k_cache is not None
    if fused:
        _need(value is not None and v_cache is not None and cache_loc is not None, "rotary_embedding: fused store args")
        _need(cache_loc.dtype == torch.int64, "rotary_embedding: cache_loc must be int64")
        _need(k_cache.dtype == _BF16 and v_cache.dtype == _BF16 and value.dtype == _BF16, "rotary_embedding: the fused store writes bf16 rows")
        v2 = value.view(T, -1)
        kc = k_cache.view(k_cache.shape[0], -1)
        vc = v_cache.view(v_cache.shape[0], -1)
        _need(kc.stride(0) == vc.stride(0), "rotary_embedding: k/v cache row strides differ")
    native.call("sgl_amd_rotary_embedding", positions.data_ptr(), q2.data_ptr(), k2.data_ptr(), cos_sin_cache.data_ptr(),
                1 if cos_sin_cache.dtype == torch.float32 else 0, T, hq, hk, head_size, rot, q2.stride(0), k2.stride(0),
                1 if is_neox else 0,
                v2.data_ptr() if fused else None, v2.stride(0) if fused else 0,
                kc.data_ptr() if fused else None, vc.data_ptr() if fused else None,
                cache_loc.data_ptr() if fused else None, kc.stride(0) if fused else 0, _stream())


# ----------------------------------------------------------------------- kv store
def store_kv_cache(k: torch.Tensor, v: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, loc: torch.Tensor,
                   num_kv_heads: Optional[int] = None, head_dim: Optional[int] = None, kv_fp8: bool = False,
                   k_scale: float = 1.0, v_scale: float = 1.0, page_size: int = 1, hnd: bool = False) -> None:
    """k_cache[loc[t]] = k[t] (and v).  With kv_fp8 / hnd the pool rows are OCP e4m3 bytes of x / scale and / or laid
    out [pages, H_kv, page_size, D] (memory_pool.py:2061-2117, 2364-2374)."""
    _dev(k, v, k_cache, v_cache, loc)
    _need(loc.dtype == torch.int64, "store_kv_cache: loc must be int64")
    T = loc.numel()
    if kv_fp8 or hnd:
        _need(num_kv_heads is not None and head_dim is not None, "store_kv_cache: fp8 / HND pools need num_kv_heads and head_dim")
        k2, v2 = k.view(T, -1), v.view(T, -1)
        _need(k2.dtype == _BF16 and v2.dtype == _BF16 and k2.shape[1] == num_kv_heads * head_dim, "store_kv_cache: bf16 [T, H_kv * D] rows")
        _need(_kv_rows_dtype_ok(k_cache, kv_fp8) and v_cache.dtype == k_cache.dtype, "store_kv_cache: pool dtype")
        native.call("sgl_amd_store_kv_cache_ex", k2.data_ptr(), v2.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), loc.data_ptr(),
                    T, num_kv_heads, head_dim, k2.stride(0), v2.stride(0), num_kv_heads * head_dim, 1 if kv_fp8 else 0,
                    float(k_scale), float(v_scale), int(page_size), 1 if hnd else 0, _stream())
        return
    k2, v2 = k.view(T, -1), v.view(T, -1)
    kc, vc = k_cache.view(k_cache.shape[0], -1), v_cache.view(v_cache.shape[0], -1)
    _need(k2.dtype == kc.dtype == _BF16 and v2.dtype == vc.dtype == _BF16, "store_kv_cache: bf16 only")
    native.call("sgl_amd_store_kv_cache", k2.data_ptr(), v2.data_ptr(), kc.data_ptr(), vc.data_ptr(), loc.data_ptr(), T,
                k2.shape[1], v2.shape[1], k2.stride(0), v2.stride(0), kc.stride(0), vc.stride(0), _stream())


# ----------------------------------------------------------------------- metadata
def create_kv_indices(req_to_token: torch.Tensor, req_pool_indices: torch.Tensor, kernel_lens: torch.Tensor,
                      kv_indptr: torch.Tensor, kv_start_idx: Optional[torch.Tensor], kv_indices: torch.Tensor) -> None:
    _dev(req_to_token, req_pool_indices, kernel_lens, kv_indptr, kv_indices)
    _need(req_to_token.dtype == torch.int32 and kernel_lens.dtype == torch.int32 and kv_indptr.dtype == torch.int32,
          "create_kv_indices: int32 req_to_token / lens / indptr")
    _need(req_pool_indices.dtype in (torch.int32, torch.int64) and kv_indices.dtype in (torch.int32, torch.int64),
          "create_kv_indices: index dtypes")
    native.call("sgl_amd_create_kv_indices", req_to_token.data_ptr(), req_to_token.stride(0), req_pool_indices.data_ptr(),
                1 if req_pool_indices.dtype == torch.int64 else 0, kernel_lens.data_ptr(), kv_indptr.data_ptr(),
                _ptr(kv_start_idx), kv_indices.data_ptr(), 1 if kv_indices.dtype == torch.int64 else 0,
                req_pool_indices.numel(), _stream())


def write_req_to_token(req_to_token: torch.Tensor, req_pool_indices: torch.Tensor, prefix_ptrs: Optional[torch.Tensor],
                       prefix_lens: torch.Tensor, seq_lens: torch.Tensor, extend_lens: torch.Tensor,
                       out_cache_loc: torch.Tensor) -> None:
    _dev(req_to_token, req_pool_indices, prefix_lens, seq_lens, extend_lens, out_cache_loc)
    for t in (req_pool_indices, prefix_lens, seq_lens, extend_lens, out_cache_loc):
        _need(t.dtype == torch.int64, "write_req_to_token: int64 metadata")
    native.call("sgl_amd_write_req_to_token", req_to_token.data_ptr(), req_to_token.stride(0), req_pool_indices.data_ptr(),
                _ptr(prefix_ptrs), prefix_lens.data_ptr(), seq_lens.data_ptr(), extend_lens.data_ptr(),
                out_cache_loc.data_ptr(), req_pool_indices.numel(), _stream())


def decode_advance(req_to_token: torch.Tensor, req_pool_indices: torch.Tensor, seq_lens: torch.Tensor,
                   new_slots: torch.Tensor, out_cache_loc: torch.Tensor) -> None:
    """One decode step's bookkeeping (page_size 1): req_to_token[pool[b], seq_lens[b]] = new_slots[b],
    out_cache_loc[b] = new_slots[b], seq_lens[b] += 1 -- one launch instead of five eager tensor ops."""
    _dev(req_to_token, req_pool_indices, seq_lens, new_slots, out_cache_loc)
    n = seq_lens.numel()
    _need(req_to_token.dtype == torch.int32 and req_to_token.dim() == 2 and req_to_token.stride(1) == 1, "decode_advance: int32 req_to_token")
    _need(req_pool_indices.dtype == torch.int64 and new_slots.dtype == torch.int64 and out_cache_loc.dtype == torch.int64
          and seq_lens.dtype == torch.int32, "decode_advance: int64 pool rows / slots, int32 seq_lens")
    _need(req_pool_indices.numel() == n and new_slots.numel() == n and out_cache_loc.numel() == n
          and all(t.is_contiguous() for t in (req_pool_indices, seq_lens, new_slots, out_cache_loc)), "decode_advance: shapes")
    native.call("sgl_amd_decode_advance", req_to_token.data_ptr(), req_to_token.stride(0), req_pool_indices.data_ptr(),
                seq_lens.data_ptr(), new_slots.data_ptr(), out_cache_loc.data_ptr(), n, _stream())


def get_last_loc(req_to_token: torch.Tensor, req_pool_indices: torch.Tensor, prefix_lens: torch.Tensor) -> torch.Tensor:
    _dev(req_to_token, req_pool_indices, prefix_lens)
    _need(req_pool_indices.dtype == torch.int64 and prefix_lens.dtype == torch.int64, "get_last_loc: int64 metadata")
    out = torch.empty_like(prefix_lens)
    native.call("sgl_amd_get_last_loc", req_to_token.data_ptr(), req_to_token.stride(0), req_pool_indices.data_ptr(),
                prefix_lens.data_ptr(), out.data_ptr(), prefix_lens.numel(), _stream())
    return out


def compute_position(extend_prefix_lens: torch.Tensor, extend_seq_lens: torch.Tensor, extend_seq_lens_sum: int
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    _dev(extend_prefix_lens, extend_seq_lens)
    _need(extend_prefix_lens.dtype == extend_seq_lens.dtype and extend_seq_lens.dtype in (torch.int32, torch.int64),
          "compute_position: lens dtype")
    positions = torch.empty(extend_seq_lens_sum, dtype=torch.int64, device=extend_seq_lens.device)
    start = torch.empty_like(extend_seq_lens)
    native.call("sgl_amd_compute_position", extend_prefix_lens.data_ptr(), extend_seq_lens.data_ptr(),
                1 if extend_seq_lens.dtype == torch.int64 else 0, positions.data_ptr(), start.data_ptr(),
                extend_seq_lens.numel(), _stream())
    return positions, start


def clamp_position(seq_lens: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev(seq_lens)
    if out is None:
        out = torch.empty(seq_lens.shape, dtype=torch.int64, device=seq_lens.device)
    native.call("sgl_amd_clamp_position", seq_lens.data_ptr(), 1 if seq_lens.dtype == torch.int64 else 0,
                out.data_ptr(), seq_lens.numel(), _stream())
    return out


def alloc_extend(prefix_lens: torch.Tensor, seq_lens: torch.Tensor, last_loc: torch.Tensor, free_pages: torch.Tensor,
                 out_indices: torch.Tensor, page_size: int) -> None:
    _dev(prefix_lens, seq_lens, last_loc, free_pages, out_indices)
    for t in (prefix_lens, seq_lens, last_loc, free_pages, out_indices):
        _need(t.dtype == torch.int64, "alloc_extend: int64 metadata")
    native.call("sgl_amd_alloc_extend", prefix_lens.data_ptr(), seq_lens.data_ptr(), last_loc.data_ptr(),
                free_pages.data_ptr(), out_indices.data_ptr(), prefix_lens.numel(), page_size, _stream())


def alloc_decode(seq_lens: torch.Tensor, last_loc: torch.Tensor, free_pages: torch.Tensor, out_indices: torch.Tensor,
                 page_size: int) -> None:
    _dev(seq_lens, last_loc, free_pages, out_indices)
    for t in (seq_lens, last_loc, free_pages, out_indices):
        _need(t.dtype == torch.int64, "alloc_decode: int64 metadata")
    native.call("sgl_amd_alloc_decode", seq_lens.data_ptr(), last_loc.data_ptr(), free_pages.data_ptr(),
                out_indices.data_ptr(), seq_lens.numel(), page_size, _stream())


# ---------------------------------------------------------------------- attention
def decode_min_chunk() -> int:
    return native.lib().sgl_amd_decode_attention_min_chunk()


def decode_workspace(batch: int, num_q_heads: int, head_dim: int, num_splits: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    acc = torch.empty((batch, num_q_heads, num_splits, head_dim), dtype=torch.float32, device=device)
    ml = torch.empty((batch, num_q_heads, num_splits, 2), dtype=torch.float32, device=device)
    return acc, ml


def decode_attention(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, out: torch.Tensor,
                     req_to_token: torch.Tensor, req_pool_indices: Optional[torch.Tensor], seq_lens: torch.Tensor,
                     sm_scale: float, num_splits: int = 1, ws_acc: Optional[torch.Tensor] = None,
                     ws_ml: Optional[torch.Tensor] = None, kv_indptr: Optional[torch.Tensor] = None,
                     flags: int = 0, batch_order: Optional[torch.Tensor] = None, kv_fp8: bool = False,
                     k_scale: float = 1.0, v_scale: float = 1.0, page_size: int = 1, hnd: bool = False,
                     sliding_window: int = -1, logit_cap: float = 0.0) -> torch.Tensor:
    """q/out [B, Hq, D]; k_cache/v_cache [slots, Hkv, D] (or [pages, Hkv, page_size, D] with hnd; uint8 e4m3 rows with
    kv_fp8); seq_lens int32 [B]."""
    _dev(q, k_cache, v_cache, out, req_to_token, seq_lens)
    B, Hq, D = q.shape
    Hkv = k_cache.shape[1]
    if kv_fp8 or hnd or sliding_window >= 0 or logit_cap > 0:
        _need(q.dtype == _BF16 and out.dtype == _BF16 and _kv_rows_dtype_ok(k_cache, kv_fp8)
              and v_cache.dtype == k_cache.dtype and k_cache.is_contiguous() and v_cache.is_contiguous(), "decode_attention: dtypes / contiguous pools")
        _need(seq_lens.dtype == torch.int32 and req_to_token.dtype == torch.int32, "decode_attention: int32 seq_lens / req_to_token")
        _need(q.stride(2) == 1 and q.stride(1) == D and out.stride(2) == 1 and out.stride(1) == D, "decode_attention: q/out head layout")
        r2t_stride = req_to_token.stride(0) if req_to_token.dim() == 2 else 0
        native.call("sgl_amd_decode_attention_ex", q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr(),
                    req_to_token.data_ptr(), r2t_stride, _ptr(req_pool_indices), seq_lens.data_ptr(), _ptr(kv_indptr),
                    B, Hq, Hkv, D, q.stride(0), out.stride(0), Hkv * D, Hkv * D, float(sm_scale), num_splits, _ptr(ws_acc),
                    _ptr(ws_ml), _ptr(batch_order), flags, 1 if kv_fp8 else 0, float(k_scale), float(v_scale), int(page_size),
                    1 if hnd else 0, int(sliding_window), float(logit_cap), _stream())
        return out
    _need(q.dtype == _BF16 and k_cache.dtype == _BF16 and v_cache.dtype == _BF16 and out.dtype == _BF16, "decode_attention: bf16 only")
    _need(seq_lens.dtype == torch.int32 and req_to_token.dtype == torch.int32, "decode_attention: int32 seq_lens / req_to_token")
    _need(req_pool_indices is None or req_pool_indices.dtype == torch.int64, "decode_attention: int64 req_pool_indices")
    _need(q.stride(2) == 1 and q.stride(1) == D and out.stride(2) == 1 and out.stride(1) == D, "decode_attention: q/out head layout")
    _need(k_cache.stride(2) == 1 and k_cache.stride(1) == D and v_cache.stride(1) == D, "decode_attention: cache layout")
    r2t_stride = req_to_token.stride(0) if req_to_token.dim() == 2 else 0
    native.call("sgl_amd_decode_attention", q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr(),
                req_to_token.data_ptr(), r2t_stride, _ptr(req_pool_indices), seq_lens.data_ptr(), _ptr(kv_indptr),
                B, Hq, Hkv, D, q.stride(0), out.stride(0), k_cache.stride(0), v_cache.stride(0), float(sm_scale),
                num_splits, _ptr(ws_acc), _ptr(ws_ml), _ptr(batch_order), flags, _stream())
    return out


def extend_attention(q: torch.Tensor, out: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                     req_to_token: torch.Tensor, req_pool_indices: torch.Tensor, seq_lens: torch.Tensor,
                     prefix_lens: torch.Tensor, qo_indptr: torch.Tensor, max_extend_len: int, sm_scale: float,
                     causal: bool = True, kv_fp8: bool = False, k_scale: float = 1.0, v_scale: float = 1.0,
                     page_size: int = 1, hnd: bool = False, sliding_window: int = -1, logit_cap: float = 0.0,
                     custom_mask: Optional[torch.Tensor] = None, mask_indptr: Optional[torch.Tensor] = None,
                     skip_prefix_custom_mask: bool = False) -> torch.Tensor:
    """q/out [T, Hq, D]; k_cache/v_cache [slots, Hkv, D] (HND / fp8 as in decode_attention); int32
    seq_lens/prefix_lens/qo_indptr; custom_mask uint8 / bool flat, mask_indptr int64 [B + 1]."""
    _dev(q, out, k_cache, v_cache, req_to_token, req_pool_indices, seq_lens, prefix_lens, qo_indptr)
    T, Hq, D = q.shape
    Hkv = k_cache.shape[1]
    if kv_fp8 or hnd or sliding_window >= 0 or logit_cap > 0 or custom_mask is not None:
        _need(q.dtype == _BF16 and out.dtype == _BF16 and _kv_rows_dtype_ok(k_cache, kv_fp8)
              and v_cache.dtype == k_cache.dtype and k_cache.is_contiguous() and v_cache.is_contiguous(), "extend_attention: dtypes / contiguous pools")
        _need(seq_lens.dtype == torch.int32 and prefix_lens.dtype == torch.int32 and qo_indptr.dtype == torch.int32, "extend_attention: int32 lens")
        _need(req_pool_indices.dtype == torch.int64 and req_to_token.dtype == torch.int32, "extend_attention: index dtypes")
        _need(q.stride(2) == 1 and q.stride(1) == D and out.stride(2) == 1 and out.stride(1) == D, "extend_attention: q/out head layout")
        if custom_mask is not None:
            _dev(custom_mask, mask_indptr)
            custom_mask = custom_mask.view(torch.uint8) if custom_mask.dtype == torch.bool else custom_mask
            _need(custom_mask.dtype == torch.uint8 and custom_mask.is_contiguous() and mask_indptr is not None
                  and mask_indptr.dtype == torch.int64, "extend_attention: uint8 custom_mask + int64 mask_indptr")
        native.call("sgl_amd_extend_attention_ex", q.data_ptr(), out.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                    req_to_token.data_ptr(), req_to_token.stride(0), req_pool_indices.data_ptr(), seq_lens.data_ptr(),
                    prefix_lens.data_ptr(), qo_indptr.data_ptr(), seq_lens.numel(), int(max_extend_len), Hq, Hkv, D,
                    q.stride(0), out.stride(0), Hkv * D, Hkv * D, float(sm_scale), 1 if causal else 0, 1 if kv_fp8 else 0,
                    float(k_scale), float(v_scale), int(page_size), 1 if hnd else 0, int(sliding_window), float(logit_cap),
                    _ptr(custom_mask), _ptr(mask_indptr), 1 if skip_prefix_custom_mask else 0, _stream())
        return out
    _need(q.dtype == _BF16 and k_cache.dtype == _BF16 and v_cache.dtype == _BF16 and out.dtype == _BF16, "extend_attention: bf16 only")
    _need(seq_lens.dtype == torch.int32 and prefix_lens.dtype == torch.int32 and qo_indptr.dtype == torch.int32,
          "extend_attention: int32 lens")
    _need(req_pool_indices.dtype == torch.int64 and req_to_token.dtype == torch.int32, "extend_attention: index dtypes")
    _need(q.stride(2) == 1 and q.stride(1) == D and out.stride(2) == 1 and out.stride(1) == D, "extend_attention: q/out head layout")
    _need(k_cache.stride(2) == 1 and k_cache.stride(1) == D and v_cache.stride(1) == D, "extend_attention: cache layout")
    native.call("sgl_amd_extend_attention", q.data_ptr(), out.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                req_to_token.data_ptr(), req_to_token.stride(0), req_pool_indices.data_ptr(), seq_lens.data_ptr(),
                prefix_lens.data_ptr(), qo_indptr.data_ptr(), seq_lens.numel(), int(max_extend_len), Hq, Hkv, D,
                q.stride(0), out.stride(0), k_cache.stride(0), v_cache.stride(0), float(sm_scale), 1 if causal else 0,
                _stream())
    return out



# ------------------------------------------------------------- shared-prefix (cascade) decode
CASCADE_MIN_SHARED = 128


class CascadeWorkspace:
    """Caller-owned buffers of the cascade decode path for batches up to `max_batch` and contexts up to
    `max_context_len`: the device plan, and the per-item partial slots shared by all layers."""

    def __init__(self, max_batch: int, num_q_heads: int, head_dim: int, max_context_len: int, device, max_items: Optional[int] = None):
        self.max_context_len = max_context_len
        self.chunk_tokens = native.lib().sgl_amd_cascade_chunk_tokens()
        chunks = (max_context_len + self.chunk_tokens - 1) // self.chunk_tokens
        # every request's private chunks + the shared chunks of at most max_batch/2 groups (x member tiles), twice over
        self.max_batch, self.max_items = max_batch, max_items or 2 * (max_batch * (chunks + 1) + (max_batch // 2 + 1) * chunks)
        self.slots = (max_context_len + self.chunk_tokens - 1) // self.chunk_tokens + 1
        self.plan = torch.zeros(native.lib().sgl_amd_cascade_plan_ints(max_batch, self.max_items), dtype=torch.int32, device=device)
        self.ws_acc = torch.empty((max_batch, num_q_heads, self.slots, head_dim), dtype=torch.float32, device=device)
        self.ws_ml = torch.empty((max_batch, num_q_heads, self.slots, 2), dtype=torch.float32, device=device)


def cascade_plan(ws: CascadeWorkspace, req_to_token: torch.Tensor, req_pool_indices: torch.Tensor, seq_lens: torch.Tensor,
                 num_q_heads: int, num_kv_heads: int, min_shared: int = CASCADE_MIN_SHARED) -> None:
    """Once per decode step: group the batch by shared KV prefix (device only, graph safe)."""
    _dev(req_to_token, req_pool_indices, seq_lens)
    B = seq_lens.numel()
    _need(B <= ws.max_batch, "cascade_plan: batch larger than the workspace")
    _need(seq_lens.dtype == torch.int32 and req_pool_indices.dtype == torch.int64 and req_to_token.dtype == torch.int32,
          "cascade_plan: dtypes")
    # NOTE: the plan / slot layout is addressed with the ACTUAL batch size B
    native.call("sgl_amd_cascade_plan", req_to_token.data_ptr(), req_to_token.stride(0), req_pool_indices.data_ptr(),
                seq_lens.data_ptr(), B, num_q_heads, num_kv_heads, min_shared, ws.max_context_len, ws.plan.data_ptr(),
                ws.max_items, _stream())


def cascade_decode_attention(ws: CascadeWorkspace, q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                             out: torch.Tensor, req_to_token: torch.Tensor, req_pool_indices: torch.Tensor,
                             seq_lens: torch.Tensor, sm_scale: float, kv_fp8: bool = False, k_scale: float = 1.0,
                             v_scale: float = 1.0, page_size: int = 1, hnd: bool = False) -> torch.Tensor:
    """Per layer, after cascade_plan(): q/out [B, Hq, D]; pools as in decode_attention (bf16 or uint8 e4m3 rows,
    [slots, Hkv, D] or [pages, Hkv, page_size, D] with hnd)."""
    _dev(q, k_cache, v_cache, out)
    B, Hq, D = q.shape
    Hkv = k_cache.shape[1]
    _need(q.dtype == _BF16 and out.dtype == _BF16 and _kv_rows_dtype_ok(k_cache, kv_fp8)
          and v_cache.dtype == k_cache.dtype, "cascade: bf16 q / out, bf16 or uint8 (e4m3) pools")
    _need(q.stride(2) == 1 and q.stride(1) == D and out.stride(2) == 1 and out.stride(1) == D, "cascade: q/out head layout")
    if kv_fp8 or hnd:
        _need(k_cache.is_contiguous() and v_cache.is_contiguous(), "cascade: contiguous pools")
        row = Hkv * D
    else:
        _need(k_cache.stride(2) == 1 and k_cache.stride(1) == D and v_cache.stride(1) == D
              and k_cache.stride(0) == v_cache.stride(0), "cascade: cache layout")
        row = k_cache.stride(0)
    native.call("sgl_amd_cascade_decode_attention_ex", q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr(),
                req_to_token.data_ptr(), req_to_token.stride(0), req_pool_indices.data_ptr(), seq_lens.data_ptr(),
                ws.plan.data_ptr(), B, ws.max_items, Hq, Hkv, D, q.stride(0), out.stride(0), row, row,
                float(sm_scale), ws.max_context_len, ws.slots, ws.ws_acc.data_ptr(), ws.ws_ml.data_ptr(),
                1 if kv_fp8 else 0, float(k_scale), float(v_scale), int(page_size), 1 if hnd else 0, _stream())
    return out


def cascade_batch_order(ws: CascadeWorkspace, batch: int) -> torch.Tensor:
    """View of the plan's batch permutation (device int32 [B]) for decode_attention(batch_order=...)."""
    off = 8 + 4 * batch + (batch + 1) + 8 * ws.max_items
    return ws.plan[off: off + batch]

def cascade_plan_summary(ws: CascadeWorkspace, batch: int) -> dict:
    """Host copy of the plan (tests / diagnostics only: this synchronises)."""
    p = ws.plan.cpu().tolist()
    B = batch
    o = 8
    req_shared = p[o:o + B]; o += B
    member_rows = p[o:o + B]; o += B
    group_qo = p[o:o + B + 1]; o += B + 1
    group_pool = p[o:o + B]; o += B
    group_kvlen = p[o:o + B]; o += B
    ng, nrows = p[1], p[2]
    order_off = o + 8 * ws.max_items
    recs = [p[o + 8 * i:o + 8 * i + 8] for i in range(p[0])]
    return dict(n_items=p[0], n_shared_items=p[3], n_groups=ng, req_shared=req_shared, member_rows=member_rows[:nrows],
                group_qo=group_qo[:ng + 1], group_pool_row=group_pool[:ng], group_kvlen=group_kvlen[:ng],
                # shared items: (group, slot, first member_rows entry, members); private: (request, slot, kv_begin, kv_n)
                items=[(r[7], r[0], r[4], r[3]) for r in recs if r[5] == 0],
                private_items=[(r[4], r[0], r[1], r[2]) for r in recs if r[5] == 1],
                batch_order=p[order_off:order_off + B])

# ----------------------------------------------------------------------- sampling
_ARGMAX_WS: dict = {}


def argmax(logits: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """torch.argmax(logits, -1) (first maximum, NaN maximal).  Decode-sized batches of wide rows are cut into column
    ranges so that the whole chip reads them (64 x 128256 bf16: 21 -> 7 us).  `out`: a contiguous int64 [B] tensor to write the
    ids into (the decode graph's own input buffer: harness/graph_runner)."""
    _dev(logits)
    _need(logits.dim() == 2 and logits.stride(1) == 1, "argmax: [B, V] row-major")
    _need(logits.dtype in (torch.float32, _BF16), "argmax: fp32 or bf16 logits")
    B, V = logits.shape
    if out is not None:
        _need(out.dtype == torch.int64 and out.shape == (B,) and out.is_contiguous() and out.device == logits.device, "argmax: out int64 [B]")
    ids = out if out is not None else torch.empty(B, dtype=torch.int64, device=logits.device)
    es = logits.element_size()
    splits = min(16, max(1, 512 // max(B, 1)))
    if (splits > 1 and V >= 16384 and B <= 65535 and V < 2 ** 32 and logits.data_ptr() % 16 == 0
            and (logits.stride(0) * es) % 16 == 0):
        import threading

        key = (logits.device, threading.get_ident(), _stream())
        ws = _ARGMAX_WS.get(key)
        need = native.lib().sgl_amd_argmax_split_workspace_bytes(B)
        if ws is None or ws.numel() < need:
            ws = _ARGMAX_WS[key] = torch.zeros(max(need, 16 * 1024), dtype=torch.uint8, device=logits.device)
        native.call("sgl_amd_argmax_split", logits.data_ptr(), 1 if logits.dtype == _BF16 else 0, ids.data_ptr(), B, V,
                    logits.stride(0), splits, ws.data_ptr(), _stream())
        return ids
    native.call("sgl_amd_argmax", logits.data_ptr(), 1 if logits.dtype == _BF16 else 0, ids.data_ptr(),
                B, V, logits.stride(0), _stream())
    return ids


def softmax_temperature_from_bf16(logits: torch.Tensor, temperatures: torch.Tensor) -> Optional[torch.Tensor]:
    """fp32 probs = softmax(logits.float() / T) straight from bf16 logits for decode-sized batches of wide rows (the widening is
    exact, the arithmetic the fp32 kernel's); None when the shape is not that case (the caller widens and uses softmax_temperature_)."""
    _dev(logits, temperatures)
    if logits.dtype != _BF16 or logits.dim() != 2 or logits.stride(1) != 1:
        return None
    B, V = logits.shape
    t = temperatures.reshape(-1)
    splits = min(64, 2048 // max(1, B))
    if not (B and splits >= 2 and V >= 4096 * splits // 8 and logits.data_ptr() % 8 == 0 and logits.stride(0) % 4 == 0 and B <= 65535
            and t.dtype == torch.float32 and t.numel() == B and t.is_contiguous()):
        return None
    probs = torch.empty((B, V + (-V) % 4), dtype=torch.float32, device=logits.device)[:, :V]     # 16-byte aligned rows
    ws = torch.empty((B * splits * 2,), dtype=torch.float32, device=logits.device)
    native.call("sgl_amd_softmax_temperature_split_bf16", logits.data_ptr(), probs.data_ptr(), t.data_ptr(), B, V, logits.stride(0),
                probs.stride(0), splits, ws.data_ptr(), _stream())
    return probs


def softmax_temperature_(logits: torch.Tensor, temperatures: torch.Tensor) -> torch.Tensor:
    _dev(logits, temperatures)
    _need(logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1, "softmax: fp32 [B, V]")
    t = temperatures.reshape(-1)
    _need(t.dtype == torch.float32 and t.numel() == logits.shape[0] and t.is_contiguous(), "softmax: temperatures [B] fp32")
    B, V = logits.shape
    splits = min(64, 2048 // max(1, B))            # column ranges per row: ~2048 workgroups over the chip
    if B and splits >= 2 and V >= 4096 * splits // 8 and logits.data_ptr() % 16 == 0 and logits.stride(0) % 4 == 0 and B <= 65535:
        # (allocated per call: under graph capture it comes from the graph's pool like every other intermediate)
        ws = torch.empty((B * splits * 2,), dtype=torch.float32, device=logits.device)
        native.call("sgl_amd_softmax_temperature_split", logits.data_ptr(), t.data_ptr(), B, V, logits.stride(0), splits,
                    ws.data_ptr(), _stream())
        return logits
    native.call("sgl_amd_softmax_temperature", logits.data_ptr(), t.data_ptr(), B, V, logits.stride(0), _stream())
    return logits



_SAMPLE_WS = {}


def sampling_lds_keep() -> int:
    return native.lib().sgl_amd_sampling_lds_keep()


def sample_ranges(batch: int, vocab: int) -> int:
    """Column ranges per row for the filtered sampler: about 512 workgroups over the chip (0 = one workgroup per row: batches that
    fill the chip on their own, or rows too short to be worth three launches)."""
    if batch <= 0 or batch > 256 or vocab < 16384:
        return 0
    r = min(16, 512 // batch)
    return r if r >= 2 else 0


def _sample_workspace(batch: int, vocab: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Caller-owned ranking workspace for nuclei larger than the LDS capacity (B*2*V words each)."""
    key = (batch, vocab, str(device))
    ws = _SAMPLE_WS.get(key)
    if ws is None:
        ws = (torch.empty((batch, 2, vocab), dtype=torch.int32, device=device),
              torch.empty((batch, 2, vocab), dtype=torch.int32, device=device))
        _SAMPLE_WS[key] = ws
    return ws


def top_k_top_p_min_p_sample(probs: torch.Tensor, top_ks: Optional[torch.Tensor], top_ps: Optional[torch.Tensor],
                             min_ps: Optional[torch.Tensor], sampling_seed: Optional[torch.Tensor],
                             positions: Optional[torch.Tensor], filtered: bool = True, use_workspace: bool = True,
                             return_n_keep: bool = False, ranges: Optional[int] = None):
    """sampler.py:567-612 / :732-750 on the gfx950 sampler.  probs fp32 [B, V] (after softmax).
    Without sampling_seed fresh per-row seeds are drawn from torch's device generator."""
    _dev(probs)
    _need(probs.dtype == torch.float32 and probs.dim() == 2 and probs.stride(1) == 1, "sample: fp32 [B, V] probs")
    B, V = probs.shape
    dev = probs.device
    if sampling_seed is None:
        sampling_seed = torch.randint(0, 2 ** 62, (B,), dtype=torch.int64, device=dev)
    seeds = sampling_seed.to(torch.int64) if sampling_seed.dtype != torch.int64 else sampling_seed
    _need(seeds.numel() == B and seeds.is_contiguous(), "sample: seeds [B]")
    if positions is not None:
        positions = positions.to(torch.int64).contiguous()
        _need(positions.numel() == B, "sample: positions [B]")
    if top_ks is not None:
        top_ks = top_ks.to(torch.int32).contiguous()
    if top_ps is not None:
        top_ps = top_ps.to(torch.float32).contiguous()
    if min_ps is not None:
        min_ps = min_ps.to(torch.float32).contiguous()
    ids = torch.empty(B, dtype=torch.int32, device=dev)
    n_keep = torch.empty(B, dtype=torch.int32, device=dev) if return_n_keep else None
    if filtered:
        ws = _sample_workspace(B, V, dev) if use_workspace else (None, None)
        # decode-sized batches of wide rows: the two full-row passes as column ranges over the whole chip, then one workgroup
        # per row on the candidate list (sampling_topk.hip: hist / collect / finish; same ids and kept counts)
        ranges = sample_ranges(B, V) if (ranges is None and use_workspace) else int(ranges or 0)
        if ranges >= 2 and probs.data_ptr() % 16 == 0 and probs.stride(0) % 4 == 0:
            key = ("ranges", B, ranges, str(dev))
            wr = _SAMPLE_WS.get(key)
            if wr is None:
                wr = _SAMPLE_WS[key] = torch.empty(native.lib().sgl_amd_sample_ranges_workspace_bytes(B, ranges) // 8 + 1,
                                                   dtype=torch.int64, device=dev)
            native.call("sgl_amd_top_k_top_p_min_p_sample_ranges", probs.data_ptr(), probs.stride(0), B, V, _ptr(top_ks), _ptr(top_ps),
                        _ptr(min_ps), seeds.data_ptr(), _ptr(positions), ids.data_ptr(), _ptr(ws[0]), _ptr(ws[1]), _ptr(n_keep),
                        ranges, wr.data_ptr(), _stream())
            return (ids, n_keep) if return_n_keep else ids
    else:           # unfiltered: 256 bytes per row of range partials (the kernel spreads decode-sized batches over the chip)
        small = torch.empty((B, 64), dtype=torch.int32, device=dev) if use_workspace else None
        ws = (small, small)
    native.call("sgl_amd_top_k_top_p_min_p_sample", probs.data_ptr(), probs.stride(0), B, V, _ptr(top_ks), _ptr(top_ps),
                _ptr(min_ps), seeds.data_ptr(), _ptr(positions), ids.data_ptr(), _ptr(ws[0]), _ptr(ws[1]),
                _ptr(n_keep), 1 if filtered else 0, _stream())
    return (ids, n_keep) if return_n_keep else ids


def sample_from_logits(logits: torch.Tensor, temperatures: torch.Tensor, top_ks: Optional[torch.Tensor],
                       top_ps: Optional[torch.Tensor], min_ps: Optional[torch.Tensor], sampling_seed: Optional[torch.Tensor],
                       positions: Optional[torch.Tensor], return_n_keep: bool = False, return_fallback: bool = False):
    """sampler.py:211-260 for the logits of a decode-sized batch -- bf16 (the model's dtype) or fp32 (what the reference's
    LogitsProcessor hands its Sampler) -- in one native call: softmax(logits / T) + the filtered sampler, the probabilities never
    written and the logits left as they are (sampling_topk.hip: candidates / finish; rows the candidates cannot decide are redone
    from their full probability row inside the finish launch).  Same ids and kept counts as softmax_temperature_{,from_bf16} +
    top_k_top_p_min_p_sample.  None when the shape is not that case (the caller takes the two calls)."""
    _dev(logits, temperatures)
    if logits.dtype not in (_BF16, torch.float32) or logits.dim() != 2 or logits.stride(1) != 1:
        return None
    B, V = logits.shape
    t = temperatures.reshape(-1)
    bf = logits.dtype == _BF16
    splits = min(64, 2048 // max(1, B))            # the softmax's own range count: the partials are its partials
    per = ((V + splits - 1) // max(1, splits) + 3) // 4 * 4
    groups = splits // 16 if (splits >= 16 and splits % 16 == 0) else 1
    if not (B and splits >= 2 and (splits <= 16 or splits % 16 == 0) and V >= 4096 * splits // 8 and groups * ((per + 1023) // 1024) <= 16
            and logits.data_ptr() % (8 if bf else 16) == 0 and logits.stride(0) % 4 == 0 and B <= 65535
            and t.dtype == torch.float32 and t.numel() == B and t.is_contiguous()):
        return None
    dev = logits.device
    if sampling_seed is None:
        sampling_seed = torch.randint(0, 2 ** 62, (B,), dtype=torch.int64, device=dev)
    seeds = sampling_seed.to(torch.int64) if sampling_seed.dtype != torch.int64 else sampling_seed
    _need(seeds.numel() == B and seeds.is_contiguous(), "sample: seeds [B]")
    if positions is not None:
        positions = positions.to(torch.int64).contiguous()
        _need(positions.numel() == B, "sample: positions [B]")
    top_ks = top_ks.to(torch.int32).contiguous() if top_ks is not None else None
    top_ps = top_ps.to(torch.float32).contiguous() if top_ps is not None else None
    min_ps = min_ps.to(torch.float32).contiguous() if min_ps is not None else None
    ids = torch.empty(B, dtype=torch.int32, device=dev)
    n_keep = torch.empty(B, dtype=torch.int32, device=dev) if return_n_keep else None
    ws = _sample_workspace(B, V, dev)
    key = ("from_logits", B, V, splits, str(dev))
    fast = _SAMPLE_WS.get(key)
    nbytes = native.lib().sgl_amd_sample_from_logits_workspace_bytes(B, splits)
    if fast is None:
        # (the probability scratch is only touched for rows redone the long way; kept with the workspace, not allocated per call)
        fast = _SAMPLE_WS[key] = (torch.empty(nbytes // 8 + 2, dtype=torch.int64, device=dev),
                                  torch.empty((B, V + (-V) % 4), dtype=torch.float32, device=dev))
    wf, scratch = fast
    native.call("sgl_amd_top_k_top_p_min_p_sample_from_logits", logits.data_ptr(), 1 if bf else 0, logits.stride(0), t.data_ptr(),
                scratch.data_ptr(), scratch.stride(0), B, V, _ptr(top_ks), _ptr(top_ps), _ptr(min_ps), seeds.data_ptr(), _ptr(positions),
                ids.data_ptr(), _ptr(ws[0]), _ptr(ws[1]), _ptr(n_keep), splits, wf.data_ptr(), _stream())
    out = (ids,)
    if return_n_keep:
        out += (n_keep,)
    if return_fallback:
        out += (wf.view(torch.int32)[nbytes // 4 - B: nbytes // 4].clone(),)
    return out[0] if len(out) == 1 else out


def sample_from_bf16_logits(logits: torch.Tensor, *args, **kwargs):
    """sample_from_logits for bf16 logits only (None otherwise)."""
    return sample_from_logits(logits, *args, **kwargs) if logits.dtype == _BF16 else None


def _renorm(probs: torch.Tensor, top_k, top_p) -> torch.Tensor:
    _dev(probs)
    probs = probs.float()
    _need(probs.dim() == 2 and probs.stride(1) == 1, "renorm: fp32 [B, V] probs")
    B, V = probs.shape
    out = torch.empty_like(probs)
    k_arr = top_k.to(torch.int32).contiguous() if isinstance(top_k, torch.Tensor) else None
    p_arr = top_p.to(torch.float32).contiguous() if isinstance(top_p, torch.Tensor) else None
    native.call("sgl_amd_top_k_top_p_renorm_probs", probs.data_ptr(), out.data_ptr(), probs.stride(0), out.stride(0), B, V,
                _ptr(k_arr), -1 if (top_k is None or k_arr is not None) else int(top_k),
                _ptr(p_arr), 2.0 if (top_p is None or p_arr is not None) else float(top_p), _stream())
    return out


def top_k_renorm_prob(probs: torch.Tensor, top_k) -> torch.Tensor:
    """sgl_kernel.top_k_renorm_prob(probs, top_k) (kernels/aot/python/sgl_kernel/sampling.py:28-72)."""
    return _renorm(probs, top_k, None)


def top_p_renorm_prob(probs: torch.Tensor, top_p) -> torch.Tensor:
    """sgl_kernel.top_p_renorm_prob(probs, top_p) (sampling.py:79-130); same result as
    sampler.py:753-762 top_p_normalize_probs_torch."""
    return _renorm(probs, None, top_p)


# ------------------------------------------------------------------- skinny / grouped GEMM
_GEMM_WS = {}
_NUM_CUS = 256          # MI355X


def skinny_gemm_max_rows() -> int:
    return native.lib().sgl_amd_skinny_gemm_max_rows()


def _gemm_workspace(device, slab_floats: int) -> torch.Tensor:
    """Split-K slabs, one buffer per (device, stream): launches on one stream are ordered, so
    consecutive GEMMs can share it."""
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    ws = _GEMM_WS.get(key)
    if ws is None or ws.numel() < slab_floats:
        ws = torch.empty(max(slab_floats, 1 << 23), dtype=torch.float32, device=device)
        _GEMM_WS[key] = ws
    return ws


def choose_gemm_config(row_blocks: int, N: int, K: int, fuse_silu: bool = False,
                       target_blocks: int = 4 * _NUM_CUS) -> Tuple[int, int]:
    """(tiles_per_wave, k_splits): aim at about four 4-wave workgroups per CU (the kernel's
    occupancy), prefer wide tiles (less activation traffic) when N alone gives enough workgroups."""
    chunks = (K + 127) // 128
    best = None
    for ntw in ((2,) if fuse_silu else (2, 1)):
        bn = 64 if fuse_silu else 64 * ntw
        tiles = max(1, row_blocks * ((N + bn - 1) // bn))
        splits = max(1, min(chunks // 2 if chunks >= 2 else 1, target_blocks // tiles, 16))
        cand = (ntw, splits, tiles * splits)
        if best is None or (best[2] < target_blocks // 2 and cand[2] > best[2]):
            best = cand
    return best[0], best[1]


def skinny_gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, fuse_silu: bool = False,
                out: Optional[torch.Tensor] = None, splits: Optional[int] = None,
                tiles_per_wave: Optional[int] = None) -> torch.Tensor:
    """F.linear(x, w, bias) for x [M <= 64, K], w [N, K] bf16 (fuse_silu: w [2N, K] -> silu(gate)*up)."""
    _dev(x, w)
    _need(x.dtype == _BF16 and w.dtype == _BF16 and x.dim() == 2 and w.dim() == 2, "skinny_gemm: bf16 2-D x / w")
    M, K = x.shape
    N = w.shape[0] // 2 if fuse_silu else w.shape[0]
    _need(w.shape[1] == K and x.stride(1) == 1 and w.stride(1) == 1, "skinny_gemm: shapes / contiguity")
    if out is None:
        out = torch.empty((M, N), dtype=_BF16, device=x.device)
    ntw_auto, splits_auto = choose_gemm_config(1, N, K, fuse_silu)
    ntw = 2 if fuse_silu else (tiles_per_wave or ntw_auto)
    if splits is None:
        splits = splits_auto
    slabs = None
    if splits > 1:
        need = native.lib().sgl_amd_skinny_gemm_slab_floats(1, N, splits, 1 if fuse_silu else 0, ntw)
        slabs = _gemm_workspace(x.device, need)
    native.call("sgl_amd_skinny_gemm", x.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(), M, N, K, x.stride(0),
                w.stride(0), out.stride(0), 1 if fuse_silu else 0, ntw, splits, _ptr(slabs), _stream())
    return out


# --------------------------------------------------------- weight-streaming decode GEMM
_WS_EPILOGUES = {"none": 0, "silu_and_mul": 1, "add_rmsnorm": 2}


def wstream_gemm_max_rows() -> int:
    return native.lib().sgl_amd_wstream_gemm_max_rows()


def wstream_supported(M: int, N: int, K: int) -> bool:
    return 0 < M <= 128 and N % 16 == 0 and K % 128 == 0 and K >= 128


# the one-tile (8 gate + 8 up rows per wave) form of the fused silu epilogue: whole rounds of 256 workgroups for Llama-3's gate_up
# (256 x 7 waves instead of 224 x 4).  Measured EQUAL to the two-tile form (8B 39.8 vs 39.5 us, 70B 160.1 vs 162.5, M = 16 37.6 vs
# 38.4: profiles/r04_exp6_gateup_interleaved.json) -- the stream sits at the chip's ~6.0 TB/s either way, the 32 idle CUs were not
# the bound -- so the policy keeps the two-tile form; the kernel form stays selectable (tiles_per_wave = 1) and tested.
WSTREAM_SILU_INTERLEAVED = False


@functools.lru_cache(maxsize=None)
def wstream_preferred(M: int, N: int, K: int) -> bool:
    """Policy on top of wstream_supported(): up to 64 rows the weight stream always beats the library GEMM
    on MI355X; at 65..128 rows it still does for the narrow projections (N <= 8192: qkv / o / down), while
    the wide ones (gate_up, lm_head) are left to hipBLASLt (benchmarks/gemm_sweep.py --M 128)."""
    return wstream_supported(M, N, K) and (M <= 64 or N <= 8192)


@functools.lru_cache(maxsize=None)
def choose_wstream_config(M: int, N: int, K: int, need_combine: bool = False, fused_silu: bool = False) -> Tuple[int, int]:
    """(waves_per_group, k_splits) for y[M,N] = x . w[N,K]^T.  One workgroup per CU is resident (its
    LDS ring holds the in-flight chunks), so the best grids are whole 256-group rounds.  The cost model
    is fitted to benchmarks/gemm_sweep.py on MI355X: a busy CU streams <= ~24 GB/s, the chip <= ~5.5 TB/s,
    ~2 us of pipeline fill, and split-K pays the fp32 partial round trip plus the combine launch."""
    tiles, nch = N // 16, K // 128
    best = None
    if fused_silu:                    # one pass, each wave owns a gate tile and its up tile: no split-K
        tiles //= 2
    wide = M > 64                     # a 128-row activation image leaves room for fewer weight waves
    for nw in (((2,) if wide else (4, 3, 2)) if fused_silu else ((5, 4) if wide else (8, 7, 6, 5, 4))):
        groups = (tiles + nw - 1) // nw
        for s in range(1, 2 if fused_silu else max(1, min(nch // 2, 32)) + 1):
            wgs = groups * s
            wg_bytes = nw * ((nch + s - 1) // s) * (8192 if fused_silu else 4096)
            full, rem = divmod(wgs, _NUM_CUS)
            t = 2.0e-6 + full * _NUM_CUS * wg_bytes / 5.5e12
            if rem:
                t += rem * wg_bytes / min(5.5e12, rem * 24e9)
            if s > 1 or (need_combine and not fused_silu):
                t += 2 * s * M * N * 4 / 4e12 + 2.5e-6
            if best is None or t < best[0]:
                best = (t, nw, s)
    return best[1], best[2]


def _x_layout(x: torch.Tensor, who: str) -> Tuple[int, int, int, int]:
    """(M, K, row stride, chunk stride) of a decode activation: row-major [M, K] (chunk stride 0) or the
    chunk-major [K/128, M, 128] form the wstream GEMMs hand to each other (blocked_activation())."""
    if x.dim() == 3:
        _need(x.shape[2] == 128 and x.stride(2) == 1, f"{who}: a chunk-major activation is [K/128, M, 128]")
        return x.shape[1], x.shape[0] * 128, x.stride(1), x.stride(0)
    _need(x.dim() == 2 and x.stride(1) == 1, f"{who}: activations are [M, K] rows or [K/128, M, 128] chunks")
    return x.shape[0], x.shape[1], x.stride(0), 0


def blocked_activation(M: int, N: int, device) -> torch.Tensor:
    """An empty chunk-major activation [N/128, M, 128] (element (m, n) at [n // 128, m, n % 128])."""
    _need(N % 128 == 0, "blocked_activation: N must be a multiple of 128")
    return torch.empty((N // 128, M, 128), dtype=_BF16, device=device)


def unblock(x: torch.Tensor) -> torch.Tensor:
    """Row-major [M, N] copy of a chunk-major activation (a no-op for a row-major one)."""
    return x.permute(1, 0, 2).reshape(x.shape[1], -1) if x.dim() == 3 else x


@functools.lru_cache(maxsize=None)
def choose_wstream_decomposition(M: int, N: int, K: int, need_combine: bool = False, fused_silu: bool = False) -> Tuple[int, int, int]:
    """(waves_per_group, tiles_per_wave, k_splits).  On top of choose_wstream_config(): a launch without split-K
    whose rows fit 64 may give every wave two output tiles (t and t + N/32) -- half the activation traffic per weight
    byte: lm_head 200 -> 174 us, the unfused gate_up 39.6 -> 39.1 (benchmarks/gemm_sweep.py --blocked).  With split-K
    the two-tile forms measured no faster (down 27.8 vs 27.5 us), so those keep one tile per wave."""
    nw, s = choose_wstream_config(M, N, K, need_combine, fused_silu)
    nch = K // 128

    def rounds_time(groups: int, wg_bytes: int, chip: float, per_cu: float) -> float:
        full, rem = divmod(groups, _NUM_CUS)
        t = 2.0e-6 + full * _NUM_CUS * wg_bytes / chip
        if rem:
            t += rem * wg_bytes / min(chip, rem * per_cu)
        return t

    if fused_silu:
        # two forms of the silu epilogue: (nw, 2) = every wave a gate tile + its up tile (N / 32 pairs to deal out), or
        # (nw, 1) = every wave ONE tile of 8 gate + 8 up rows (N / 16 tiles).  The second one exists for the widths whose pairs
        # do not make whole rounds of the chip: Llama-3-8B gate_up has 896 pairs = 224 workgroups of 4 waves (32 CUs idle)
        # but 1792 tiles = 256 workgroups of 7 waves (measured equal: see WSTREAM_SILU_INTERLEAVED)
        best = (rounds_time((N // 32 + nw - 1) // nw, nw * nch * 8192, 6.0e12, 26e9), nw, 2)
        if WSTREAM_SILU_INTERLEAVED and M <= 64 and N % 16 == 0:
            for nw1 in (8, 7, 6, 5, 4):
                t1 = rounds_time((N // 16 + nw1 - 1) // nw1, nw1 * nch * 4096, 6.4e12, 26e9)
                if t1 < best[0] * 0.97:
                    best = (t1, nw1, 1)
        return best[1], best[2], s
    if s != 1 or M > 64 or N % 32 != 0:
        return nw, 1, s

    one = rounds_time((N // 16 + nw - 1) // nw, nw * nch * 4096, 5.3e12, 24e9)
    best = (one, nw, 1)
    for nw2 in (4, 3, 2):
        two = rounds_time((N // 32 + nw2 - 1) // nw2, nw2 * nch * 8192, 6.0e12, 26e9)
        if two < best[0] * 0.98:
            best = (two, nw2, 2)
    return best[1], best[2], 1


def wstream_gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: str = "none",
                 residual: Optional[torch.Tensor] = None, norm_weight: Optional[torch.Tensor] = None, eps: float = 0.0,
                 out: Optional[torch.Tensor] = None, waves_per_group: Optional[int] = None,
                 splits: Optional[int] = None, out_blocked: bool = False, tiles_per_wave: Optional[int] = None) -> torch.Tensor:
    """Decode-batch F.linear(x, w, bias) on the weight-streaming kernel, optionally followed (in the
    split-K combine kernel) by silu_and_mul or by fused_add_rmsnorm(out, residual, norm_weight, eps).
    x may be chunk-major (see _x_layout); out_blocked=True returns the result chunk-major."""
    _dev(x, w)
    _need(x.dtype == _BF16 and w.dtype == _BF16 and w.dim() == 2, "wstream_gemm: bf16 x / 2-D bf16 w")
    M, K, x_rs, x_cs = _x_layout(x, "wstream_gemm")
    N = w.shape[0]
    _need(w.shape[1] == K and w.stride(1) == 1, "wstream_gemm: shapes / contiguity")
    ep = _WS_EPILOGUES[epilogue]
    n_out = N // 2 if ep == 1 else N
    if out is None:
        out = blocked_activation(M, n_out, x.device) if out_blocked else torch.empty((M, n_out), dtype=_BF16, device=x.device)
    _need(out.dtype == _BF16, "wstream_gemm: bf16 out")
    oM, oN, y_rs, y_cs = _x_layout(out, "wstream_gemm (out)")
    _need((oM, oN) == (M, n_out), "wstream_gemm: out shape")
    if ep == 2:
        _need(residual is not None and norm_weight is not None and residual.shape == (M, N) and residual.stride(1) == 1
              and residual.dtype == _BF16 and norm_weight.dtype == _BF16, "wstream_gemm: add_rmsnorm needs residual [M,N] and norm_weight")
    tpw_auto = 1
    if waves_per_group is None or splits is None:
        # silu_and_mul: one pass with two tiles per wave when the caller does not force a split
        one_pass = ep == 1 and splits in (None, 1) and (N % 32 == 0 or (tiles_per_wave == 1 and N % 16 == 0)) and bias is None
        if waves_per_group is None and splits is None and tiles_per_wave is None:
            nw_auto, tpw_auto, s_auto = choose_wstream_decomposition(M, N, K, ep != 0, one_pass)
        else:
            nw_auto, s_auto = choose_wstream_config(M, N, K, ep != 0, one_pass)
    nw = waves_per_group or nw_auto
    s = splits or s_auto
    one_pass = ep == 1 and s == 1 and (N % 32 == 0 or (tiles_per_wave == 1 and N % 16 == 0)) and bias is None
    tpw = (tiles_per_wave or (tpw_auto if (waves_per_group is None and splits is None) else 2)) if one_pass else (tiles_per_wave or tpw_auto)
    # what is launched is decided by the C side from the values PASSED (wstream_gemm.hip `fused_silu`): restated here with the final
    # tiles-per-wave, so that the workspace below exists exactly when the launch has a combine pass -- a silu GEMM whose N is a
    # multiple of 16 but not of 32 runs the interleaved one-pass form when the automatic decomposition says one tile per wave
    one_pass = ep == 1 and s == 1 and bias is None and (N % 16 == 0 if tpw == 1 else N % 32 == 0)
    ws = _gemm_workspace(x.device, native.lib().sgl_amd_wstream_gemm_workspace_floats(M, N, s)) if ((s > 1 or ep) and not one_pass) else None
    native.call("sgl_amd_wstream_gemm", x.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(), M, N, K, x_rs, x_cs,
                w.stride(0), y_rs, y_cs, ep, _ptr(residual), residual.stride(0) if residual is not None else 0,
                _ptr(norm_weight), float(eps), nw, tpw, s, _ptr(ws), _stream())
    return out


def wstream_qkv_rope(x: torch.Tensor, w_qkv: torch.Tensor, bias: Optional[torch.Tensor], positions: torch.Tensor,
                     cos_sin_cache: torch.Tensor, num_q_heads: int, num_kv_heads: int, head_dim: int,
                     k_cache: torch.Tensor, v_cache: torch.Tensor, cache_loc: torch.Tensor,
                     waves_per_group: Optional[int] = None, splits: Optional[int] = None,
                     tiles_per_wave: Optional[int] = None, kv_fp8: bool = False, k_scale: float = 1.0,
                     v_scale: float = 1.0, page_size: int = 1, hnd: bool = False) -> torch.Tensor:
    """Decode-batch qkv_proj + neox rotary embedding + KV-pool store (one GEMM + combine pair): returns the
    rotated q [M, Hq*D]; the rotated k rows and the v rows land in k_cache / v_cache at cache_loc, in the pool's
    format (kv_fp8 / hnd as in store_kv_cache)."""
    _dev(x, w_qkv, positions, cos_sin_cache, k_cache, v_cache, cache_loc)
    M, K, x_rs, x_cs = _x_layout(x, "wstream_qkv_rope")
    N = (num_q_heads + 2 * num_kv_heads) * head_dim
    _need(x.dtype == _BF16 and w_qkv.dtype == _BF16 and w_qkv.shape == (N, K) and w_qkv.stride(1) == 1,
          "wstream_qkv_rope: x [M,K] / w_qkv [(Hq+2Hkv)*D, K] bf16")
    _need(positions.dtype == torch.int64 and cache_loc.dtype == torch.int64 and positions.numel() == M and cache_loc.numel() == M,
          "wstream_qkv_rope: int64 positions / cache_loc of length M")
    _need(cos_sin_cache.dtype in (_BF16, torch.float32) and cos_sin_cache.is_contiguous() and cos_sin_cache.shape[-1] == head_dim,
          "wstream_qkv_rope: cos_sin_cache [max_pos, head_dim]")
    row = num_kv_heads * head_dim
    if kv_fp8 or hnd:
        _need(_kv_rows_dtype_ok(k_cache, kv_fp8) and v_cache.dtype == k_cache.dtype and k_cache.is_contiguous()
              and v_cache.is_contiguous() and k_cache.numel() % row == 0 and v_cache.numel() == k_cache.numel(),
              "wstream_qkv_rope: contiguous fp8 (uint8) / HND pool of Hkv*D rows")
        kc, vc, cache_rs = k_cache, v_cache, row
    else:
        kc, vc = k_cache.view(k_cache.shape[0], -1), v_cache.view(v_cache.shape[0], -1)
        _need(kc.dtype == _BF16 and vc.dtype == _BF16 and kc.stride(0) == vc.stride(0) and kc.shape[1] == row,
              "wstream_qkv_rope: bf16 KV pool rows of Hkv*D")
        cache_rs = kc.stride(0)
    if waves_per_group is None or splits is None:
        nw_auto, s_auto = choose_wstream_config(M, N, K, True)
    nw, s = waves_per_group or nw_auto, splits or s_auto
    q_out = torch.empty((M, num_q_heads * head_dim), dtype=_BF16, device=x.device)
    ws = _gemm_workspace(x.device, native.lib().sgl_amd_wstream_gemm_workspace_floats(M, N, s))
    native.call("sgl_amd_wstream_qkv_rope", x.data_ptr(), w_qkv.data_ptr(), _ptr(bias), q_out.data_ptr(), M, K, num_q_heads,
                num_kv_heads, head_dim, x_rs, x_cs, w_qkv.stride(0), q_out.stride(0), positions.data_ptr(),
                cos_sin_cache.data_ptr(), 1 if cos_sin_cache.dtype == torch.float32 else 0, cos_sin_cache.shape[-1],
                kc.data_ptr(), vc.data_ptr(), cache_loc.data_ptr(), cache_rs, 1 if kv_fp8 else 0, float(k_scale), float(v_scale),
                int(page_size), 1 if hnd else 0, nw, tiles_per_wave or 1, s, ws.data_ptr(), _stream())
    return q_out


def moe_wstream_gemm(a: torch.Tensor, w: torch.Tensor, c: torch.Tensor, sorted_token_ids: torch.Tensor,
                     expert_ids: torch.Tensor, num_tokens_post_padded: torch.Tensor,
                     topk_weights: Optional[torch.Tensor], mul_routed_weight: bool, top_k_div: int, num_valid_ids: int,
                     block_m: int, fuse_silu: bool = False, round_before_scale: bool = False,
                     waves_per_group: Optional[int] = None) -> torch.Tensor:
    """moe_grouped_gemm on the weight-streaming kernel (no split-K): a [rows, K] bf16, w [E, N(or 2N), K] bf16,
    c [num_valid_ids, N] bf16 or fp32; needs N % 16 == 0 and K % 128 == 0."""
    _dev(a, w, c, sorted_token_ids, expert_ids, num_tokens_post_padded)
    _need(a.dtype == _BF16 and w.dtype == _BF16 and w.dim() == 3, "moe_wstream_gemm: bf16 a / w[E,N,K]")
    _need(c.dtype in (_BF16, torch.float32) and c.dim() == 2, "moe_wstream_gemm: c bf16 / fp32 [rows, N]")
    _need(sorted_token_ids.dtype == torch.int32 and expert_ids.dtype == torch.int32
          and num_tokens_post_padded.dtype == torch.int32, "moe_wstream_gemm: int32 metadata")
    E, WN, K = w.shape
    N = WN // 2 if fuse_silu else WN
    _need(a.shape[1] == K and c.shape[1] == N and a.stride(1) == 1 and w.stride(2) == 1 and c.stride(1) == 1,
          "moe_wstream_gemm: shapes / contiguity")
    max_m_blocks = expert_ids.numel()
    _need(sorted_token_ids.numel() >= max_m_blocks * block_m, "moe_wstream_gemm: sorted_token_ids too short")
    if mul_routed_weight:
        _need(topk_weights is not None and topk_weights.dtype == torch.float32 and topk_weights.is_contiguous(),
              "moe_wstream_gemm: fp32 topk_weights")
    if waves_per_group is None:
        # widest group that still gives every CU a workgroup (about num_valid_ids / block_m + E row blocks are live)
        live = max(1, min(max_m_blocks, num_valid_ids // block_m + E))
        tiles = (N // 16)
        cands = (4, 3, 2) if fuse_silu else (8, 7, 6, 5, 4)
        waves_per_group = next((nw for nw in cands if -(-tiles // nw) * live >= _NUM_CUS), cands[-1])
    native.call("sgl_amd_wstream_moe_gemm", a.data_ptr(), w.data_ptr(), c.data_ptr(), sorted_token_ids.data_ptr(),
                expert_ids.data_ptr(), num_tokens_post_padded.data_ptr(), _ptr(topk_weights),
                1 if mul_routed_weight else 0, 1 if round_before_scale else 0, top_k_div, num_valid_ids, N, K,
                a.stride(0), w.stride(1), w.stride(0), c.stride(0), block_m, max_m_blocks, 1 if fuse_silu else 0,
                1 if c.dtype == torch.float32 else 0, waves_per_group, _stream())
    return c


def moe_tiled_gemm_block_m() -> int:
    return native.lib().sgl_amd_moe_tiled_gemm_block_m()


def moe_tiled_gemm(a: torch.Tensor, w: torch.Tensor, c: torch.Tensor, sorted_token_ids: torch.Tensor,
                   expert_ids: torch.Tensor, num_tokens_post_padded: torch.Tensor, topk_weights: Optional[torch.Tensor],
                   mul_routed_weight: bool, top_k_div: int, num_valid_ids: int, block_m: int, fuse_silu: bool = False,
                   round_before_scale: bool = False, tile_rows: Optional[int] = None) -> torch.Tensor:
    """moe_grouped_gemm on the row-tiled MFMA kernels (prefill-sized batches): block_m = the align block size (128 or
    256), tile_rows = rows of an expert per workgroup tile (128, or 256 over a 256 alignment: the 256 x 256 x 64 form);
    a [rows, K] bf16, w [E, N(or 2N), K] bf16, c [num_valid_ids, N] bf16 or fp32."""
    _dev(a, w, c, sorted_token_ids, expert_ids, num_tokens_post_padded)
    tile_rows = block_m if tile_rows is None else tile_rows
    _need(block_m in (128, 256) and tile_rows in (128, 256) and tile_rows <= block_m, "moe_tiled_gemm: block_m / tile_rows in (128, 256)")
    _need(a.dtype == _BF16 and w.dtype == _BF16 and w.dim() == 3 and c.dtype in (_BF16, torch.float32) and c.dim() == 2,
          "moe_tiled_gemm: bf16 a / w[E,N,K], c bf16 / fp32")
    _need(sorted_token_ids.dtype == torch.int32 and expert_ids.dtype == torch.int32
          and num_tokens_post_padded.dtype == torch.int32, "moe_tiled_gemm: int32 metadata")
    E, WN, K = w.shape
    N = WN // 2 if fuse_silu else WN
    _need(a.shape[1] == K and c.shape[1] == N and a.stride(1) == 1 and w.stride(2) == 1 and c.stride(1) == 1,
          "moe_tiled_gemm: shapes / contiguity")
    max_m_blocks = expert_ids.numel()
    _need(sorted_token_ids.numel() >= max_m_blocks * block_m, "moe_tiled_gemm: sorted_token_ids too short")
    if mul_routed_weight:
        _need(topk_weights is not None and topk_weights.dtype == torch.float32 and topk_weights.is_contiguous(),
              "moe_tiled_gemm: fp32 topk_weights")
    native.call("sgl_amd_moe_tiled_gemm_ex", a.data_ptr(), w.data_ptr(), c.data_ptr(), sorted_token_ids.data_ptr(),
                expert_ids.data_ptr(), num_tokens_post_padded.data_ptr(), _ptr(topk_weights), 1 if mul_routed_weight else 0,
                1 if round_before_scale else 0, top_k_div, num_valid_ids, N, K, a.stride(0), w.stride(1), w.stride(0),
                c.stride(0), max_m_blocks, 1 if fuse_silu else 0, 1 if c.dtype == torch.float32 else 0, int(block_m), int(tile_rows),
                _stream())
    return c


MOE_TILE_OVERRIDE = None                # tests / benchmarks: (align block, up tile rows, down tile rows)
MOE_TILED_MIN_ROWS_PER_EXPERT = 96     # above this an expert fills most of a 128-row tile: the MFMA-bound form wins
MOE_TILE256_MIN_ROWS_PER_EXPERT = 768  # above this the 256-row alignment's padding (128 rows per expert on average) is
                                       # worth the 256 x 256 x 64 form's rate


def moe_tile_plan(numel: int, num_experts: int, n_up: int, n_down: int):
    """(align block, tile rows of the up projection, tile rows of the down projection) for a prefill-sized batch of
    `numel` (token, k) pairs: the 256 x 256 x 64 form once an expert owns enough rows that the 256-row alignment's padding
    (128 rows per expert on average) costs less than the form gains (measured on Mixtral's shapes, 4096 tokens = 1024 rows
    per expert: up projection 977 vs 714 TF/s, down projection 820-870 vs 715 with its 2.25 rounds of workgroups;
    profiles/r04_exp4_moe_gemm_ab.json)."""
    rows = numel // max(1, num_experts)
    if rows < MOE_TILE256_MIN_ROWS_PER_EXPERT:
        return 128, 128, 128
    return 256, 256, 256


def moe_grouped_gemm(a: torch.Tensor, w: torch.Tensor, c: torch.Tensor, sorted_token_ids: torch.Tensor,
                     expert_ids: torch.Tensor, num_tokens_post_padded: torch.Tensor,
                     topk_weights: Optional[torch.Tensor], mul_routed_weight: bool, top_k_div: int, num_valid_ids: int,
                     block_m: int, fuse_silu: bool = False, round_before_scale: bool = False,
                     splits: Optional[int] = None, tiles_per_wave: Optional[int] = None) -> torch.Tensor:
    """invoke_fused_moe_kernel equivalent (fused_moe_triton_kernels.py:771): a [rows, K] bf16,
    w [E, N(or 2N), K] bf16, c [num_valid_ids, N] bf16 or fp32."""
    _dev(a, w, c, sorted_token_ids, expert_ids, num_tokens_post_padded)
    _need(a.dtype == _BF16 and w.dtype == _BF16 and w.dim() == 3, "moe_grouped_gemm: bf16 a / w[E,N,K]")
    _need(c.dtype in (_BF16, torch.float32) and c.dim() == 2, "moe_grouped_gemm: c bf16 / fp32 [rows, N]")
    _need(sorted_token_ids.dtype == torch.int32 and expert_ids.dtype == torch.int32
          and num_tokens_post_padded.dtype == torch.int32, "moe_grouped_gemm: int32 metadata")
    E, WN, K = w.shape
    N = WN // 2 if fuse_silu else WN
    _need(a.shape[1] == K and c.shape[1] == N and a.stride(1) == 1 and w.stride(2) == 1 and c.stride(1) == 1,
          "moe_grouped_gemm: shapes / contiguity")
    max_m_blocks = expert_ids.numel()
    _need(sorted_token_ids.numel() >= max_m_blocks * block_m, "moe_grouped_gemm: sorted_token_ids too short")
    if mul_routed_weight:
        _need(topk_weights is not None and topk_weights.dtype == torch.float32 and topk_weights.is_contiguous(),
              "moe_grouped_gemm: fp32 topk_weights")
    # only about num_valid_ids / block_m + E of the row blocks are live
    live = max(1, min(max_m_blocks, num_valid_ids // block_m + E))
    ntw_auto, splits_auto = choose_gemm_config(live, N, K, fuse_silu)
    ntw = 2 if fuse_silu else (tiles_per_wave or ntw_auto)
    if splits is None:
        splits = splits_auto
    slabs = None
    if splits > 1:
        need = native.lib().sgl_amd_skinny_gemm_slab_floats(max_m_blocks, N, splits, 1 if fuse_silu else 0, ntw)
        slabs = _gemm_workspace(a.device, need)
    native.call("sgl_amd_moe_grouped_gemm", a.data_ptr(), w.data_ptr(), c.data_ptr(), sorted_token_ids.data_ptr(),
                expert_ids.data_ptr(), num_tokens_post_padded.data_ptr(), _ptr(topk_weights),
                1 if mul_routed_weight else 0, 1 if round_before_scale else 0, top_k_div, num_valid_ids, N, K, E,
                a.stride(0), w.stride(1), w.stride(0), c.stride(0), block_m, max_m_blocks, 1 if fuse_silu else 0,
                1 if c.dtype == torch.float32 else 0, ntw, splits, _ptr(slabs), _stream())
    return c


# ---------------------------------------------------------------------------- MoE
def topk_softmax(gating_output: torch.Tensor, topk: int, renormalize: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """sgl_kernel.topk_softmax / fused_topk (topk.py:827): fp32 weights [M,k], int32 ids [M,k]."""
    _dev(gating_output)
    _need(gating_output.dim() == 2 and gating_output.stride(1) == 1 and gating_output.dtype in (_BF16, torch.float32),
          "topk_softmax: [M, E] fp32 / bf16 gate logits")
    M, E = gating_output.shape
    w = torch.empty((M, topk), dtype=torch.float32, device=gating_output.device)
    ids = torch.empty((M, topk), dtype=torch.int32, device=gating_output.device)
    native.call("sgl_amd_topk_softmax", gating_output.data_ptr(), 1 if gating_output.dtype == _BF16 else 0, w.data_ptr(),
                ids.data_ptr(), M, E, topk, gating_output.stride(0), 1 if renormalize else 0, _stream())
    return w, ids


def moe_align_block_size(topk_ids: torch.Tensor, block_size: int, num_experts: int
                         ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """moe_runner/triton_utils/moe_align_block_size.py:31-170: (sorted_token_ids, expert_ids, num_tokens_post_padded)."""
    _dev(topk_ids)
    _need(topk_ids.dtype in (torch.int32, torch.int64) and topk_ids.is_contiguous(), "moe_align_block_size: int ids")
    numel = topk_ids.numel()
    if numel < num_experts + 1:
        max_padded = numel * block_size
    else:
        max_padded = numel + (num_experts + 1) * (block_size - 1)
    dev = topk_ids.device
    max_blocks = (max_padded + block_size - 1) // block_size
    max_padded = max_blocks * block_size          # whole row blocks, so the grouped GEMM never reads past the end
    sorted_ids = torch.empty(max_padded, dtype=torch.int32, device=dev)
    expert_ids = torch.empty(max_blocks, dtype=torch.int32, device=dev)
    post = torch.empty(1, dtype=torch.int32, device=dev)
    native.call("sgl_amd_moe_align_block_size", topk_ids.data_ptr(), 1 if topk_ids.dtype == torch.int64 else 0, numel,
                num_experts, block_size, sorted_ids.data_ptr(), expert_ids.data_ptr(), post.data_ptr(), max_padded,
                max_blocks, _stream())
    return sorted_ids, expert_ids, post


def moe_sum_reduce(x: torch.Tensor, out: torch.Tensor, routed_scaling_factor: float = 1.0) -> torch.Tensor:
    """sgl_kernel.moe_sum_reduce: x [M, topk, H] (bf16 or fp32) -> out [M, H] bf16."""
    _dev(x, out)
    _need(x.dim() == 3 and x.stride(2) == 1 and x.dtype in (_BF16, torch.float32), "moe_sum_reduce: [M, k, H] input")
    _need(out.dtype == _BF16 and out.dim() == 2 and out.stride(1) == 1, "moe_sum_reduce: bf16 [M, H] output")
    M, k, H = x.shape
    native.call("sgl_amd_moe_sum_reduce", x.data_ptr(), 1 if x.dtype == torch.float32 else 0, out.data_ptr(), M, k, H,
                x.stride(0), x.stride(1), out.stride(0), float(routed_scaling_factor), _stream())
    return out


def choose_moe_block_m(num_pairs: int, num_experts: int) -> int:
    """Row-block height of the grouped GEMM: the smallest of 16/32/48/64 that holds 1.5x an average expert's
    rows -- an expert whose rows spill into a second block streams its weights twice, padded rows cost nothing
    (the kernel is bound by the weight stream)."""
    avg = (num_pairs + num_experts - 1) // max(1, num_experts)
    for bm in (16, 32, 48):
        if avg + avg // 2 <= bm:
            return bm
    return 64


def fused_experts(hidden_states: torch.Tensor, w13: torch.Tensor, w2: torch.Tensor, topk_weights: torch.Tensor,
                  topk_ids: torch.Tensor, routed_scaling_factor: float = 1.0, out: Optional[torch.Tensor] = None
                  ) -> torch.Tensor:
    """fused_experts (moe_runner/triton_utils/fused_moe.py:242-455) on the gfx950 kernels:
    align -> grouped up-GEMM with silu(gate)*up epilogue -> grouped down-GEMM x router weight (fp32)
    -> sum over top-k.  hidden [M, K] bf16, w13 [E, 2N, K], w2 [E, K, N]."""
    M, K = hidden_states.shape
    E, N2, _ = w13.shape
    N = N2 // 2
    topk = topk_ids.shape[1]
    numel = M * topk
    dev = hidden_states.device
    if out is None:
        out = torch.empty((M, K), dtype=_BF16, device=dev)
    if M == 0:
        return out
    if numel // E >= MOE_TILED_MIN_ROWS_PER_EXPERT and K % 64 == 0 and N % 64 == 0:
        # prefill-sized batch: row-tiled MFMA kernels, every expert's weights read once per 128 / 256 of its rows
        block_m, up_rows, down_rows = moe_tile_plan(numel, E, N, K) if K % 64 == 0 else (128, 128, 128)
        if MOE_TILE_OVERRIDE is not None:
            block_m, up_rows, down_rows = MOE_TILE_OVERRIDE
        sorted_ids, expert_ids, post = moe_align_block_size(topk_ids, block_m, E)
        inter = torch.empty((numel, N), dtype=_BF16, device=dev)
        moe_tiled_gemm(hidden_states, w13, inter, sorted_ids, expert_ids, post, None, False, topk, numel, block_m, fuse_silu=True,
                       tile_rows=up_rows)
        down = torch.empty((numel, K), dtype=torch.float32, device=dev)
        moe_tiled_gemm(inter, w2, down, sorted_ids, expert_ids, post, topk_weights.reshape(-1).contiguous(), True, 1, numel,
                       block_m, round_before_scale=True, tile_rows=down_rows)
        moe_sum_reduce(down.view(M, topk, K), out, routed_scaling_factor)
        return out
    block_m = choose_moe_block_m(numel, E)
    sorted_ids, expert_ids, post = moe_align_block_size(topk_ids, block_m, E)
    inter = torch.empty((numel, N), dtype=_BF16, device=dev)
    # the weight-streaming form wherever its tiling fits (N % 16, K % 128); the skinny kernel takes ragged shapes
    up = moe_wstream_gemm if (N % 16 == 0 and K % 128 == 0) else moe_grouped_gemm
    dn = moe_wstream_gemm if (K % 16 == 0 and N % 128 == 0) else moe_grouped_gemm
    up(hidden_states, w13, inter, sorted_ids, expert_ids, post, None, False, topk, numel, block_m, fuse_silu=True)
    down = torch.empty((numel, K), dtype=torch.float32, device=dev)
    dn(inter, w2, down, sorted_ids, expert_ids, post, topk_weights.reshape(-1).contiguous(), True, 1, numel, block_m,
       round_before_scale=True)
    moe_sum_reduce(down.view(M, topk, K), out, routed_scaling_factor)
    return out

# -------------------------------------------------------------------------- probe
def probe_mfma_16x16x32(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _dev(a, b)
    _need(a.shape == (16, 32) and b.shape == (32, 16) and a.dtype == _BF16 and b.dtype == _BF16, "probe shapes")
    c = torch.empty((16, 16), dtype=torch.float32, device=a.device)
    native.call("sgl_amd_probe_mfma_16x16x32", a.contiguous().data_ptr(), b.contiguous().data_ptr(), c.data_ptr(), _stream())
    return c
