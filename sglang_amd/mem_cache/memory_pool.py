"""Request->token and token->KV pools (device memory laid out for one MI355X).

Mirrors /root/reference/python/sglang/srt/mem_cache/memory_pool.py:
  ReqToTokenPool      :256-333   int32 [size+1, max_context_len], row 0 = padding row
  MHATokenToKVPool    :1759-2456 per-layer K/V [size+page_size, H_kv, D], slot 0's page = sink

HBM layout: one contiguous allocation per K and per V, shaped
[layers, size+page_size, H_kv, D] (NHD rows of H_kv*D*2 bytes); the per-layer
buffers handed to the attention backend are views of it.  A KV row of one
token and one kv head is D*2 contiguous bytes -- the unit the attention
kernels gather with 16-byte lanes.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch


def _host_scale(layer, name: str) -> float:
    """The layer's fp8 KV scale as a HOST float: `<name>_float` (radix_attention.py:129-130), else `<name>` when it is a
    Python number, else 1.0 when the layer carries no scale at all.  A scale that exists only as a TENSOR raises: reading
    it would synchronise (and fail inside a graph capture), ignoring it would silently store and dequantise with 1.0."""
    v = getattr(layer, name + "_float", None)
    if v is None:
        v = getattr(layer, name, None)
        if v is None:
            return 1.0
        if isinstance(v, torch.Tensor):
            raise ValueError(f"fp8 KV pool: the layer's {name} exists only as a tensor; set {name}_float (a host float) when "
                             f"the checkpoint scales are loaded (RadixAttention.set_kv_scales)")
    return float(v)


@dataclass
class KVWriteLoc:
    """memory_pool.py KVWriteLoc(loc, swa_loc): where this step's K/V rows go."""

    loc: torch.Tensor
    swa_loc: Optional[torch.Tensor] = None


def unwrap_write_loc(loc_info):
    if isinstance(loc_info, KVWriteLoc) or (not isinstance(loc_info, torch.Tensor) and hasattr(loc_info, "loc")):
        return loc_info.loc, getattr(loc_info, "swa_loc", None), None       # (any KVWriteLoc-shaped record, the reference's too)
    return loc_info, None, None


class ReqToTokenPool:
    """Maps a request slot to the KV slots of its tokens."""

    def __init__(self, size: int, max_context_len: int, device, enable_memory_saver: bool = False):
        self.size = size
        self._alloc_size = size + 1   # row 0 absorbs padded (graph) batch entries
        self.max_context_len = max_context_len
        self.device = device
        self.req_to_token = torch.zeros((self._alloc_size, max_context_len), dtype=torch.int32, device=device)
        self.free_slots: List[int] = list(range(1, self._alloc_size))

    def write(self, indices, values) -> None:
        self.req_to_token[indices] = values

    def available_size(self) -> int:
        return len(self.free_slots)

    def alloc(self, reqs) -> Optional[List[int]]:
        """Assign `req_pool_idx` to every request that has none (memory_pool.py:292-324)."""
        need = [r for r in reqs if getattr(r, "req_pool_idx", None) is None]
        if len(need) > len(self.free_slots):
            return None
        if need:
            picked = self.free_slots[-len(need):]
            del self.free_slots[-len(need):]
            for r, idx in zip(need, picked):
                r.req_pool_idx = idx
        return [r.req_pool_idx for r in reqs]

    def free(self, req) -> None:
        assert req.req_pool_idx is not None, "request must have req_pool_idx"
        self.free_slots.append(req.req_pool_idx)
        req.req_pool_idx = None

    def clear(self) -> None:
        self.free_slots = list(range(1, self._alloc_size))


FP8_E4M3 = torch.float8_e4m3fn


class MHATokenToKVPool:
    """Multi-head K/V cache (memory_pool.py:1759-2456): bf16 or OCP e4m3 rows ("fp8_e4m3": K / k_scale and
    V / v_scale, :2364-2374), NHD [slots, H_kv, D] or -- `use_hnd` with page_size > 1 -- HND
    [pages, H_kv, page_size, D] (:2061-2117).  fp8 rows are stored as uint8 (`store_dtype`, :1800-1806)."""

    def __init__(self, size: int, page_size: int, dtype: torch.dtype, head_num: int, head_dim: int, layer_num: int,
                 device, enable_memory_saver: bool = False, v_head_dim: Optional[int] = None,
                 start_layer: Optional[int] = None, end_layer: Optional[int] = None, use_hnd: bool = False):
        assert dtype in (torch.bfloat16, FP8_E4M3), "the gfx950 KV kernels read bf16 or OCP e4m3 rows"
        self.size = size
        self.page_size = page_size
        self.dtype = dtype
        self.is_fp8 = dtype == FP8_E4M3
        self.store_dtype = torch.uint8 if self.is_fp8 else dtype
        self.device = device
        self.head_num = head_num
        self.head_dim = head_dim
        self.v_head_dim = v_head_dim if v_head_dim is not None else head_dim
        assert self.v_head_dim == head_dim, "K and V rows share one layout on this path"
        self.layer_num = layer_num
        self.start_layer = start_layer or 0
        self.end_layer = end_layer if end_layer is not None else layer_num - 1
        self.use_hnd = bool(use_hnd) and page_size > 1          # at page_size 1 the two layouts coincide
        rows = size + page_size
        if self.use_hnd:
            assert page_size & (page_size - 1) == 0, "HND pools need a power-of-two page_size"
            shape = (layer_num, rows // page_size, head_num, page_size, head_dim)
        else:
            shape = (layer_num, rows, head_num, head_dim)
        # zero-filled so the sink page never holds NaN patterns (0x00 is +0 in e4m3 too)
        self._k_all = torch.zeros(shape, dtype=self.store_dtype, device=device)
        self._v_all = torch.zeros(shape, dtype=self.store_dtype, device=device)
        self.k_buffer = [self._k_all[i] for i in range(layer_num)]
        self.v_buffer = [self._v_all[i] for i in range(layer_num)]
        self.row_dim = head_num * head_dim
        self.v_row_dim = head_num * self.v_head_dim

    @property
    def slot_stride(self) -> int:
        """Elements between consecutive slots of an NHD pool (H_kv * D); unused by the HND formula."""
        return self.head_num * self.head_dim

    def kernel_format(self, layer=None):
        """(kv_fp8, k_scale, v_scale, page_size, hnd) for the attention / store kernels.  Scales are the layer's
        (RadixAttention.k_scale / v_scale, loaded from the checkpoint) or 1.0 (memory_pool.py:2364-2369)."""
        # host floats only (radix_attention.py:129-130): float(k_scale tensor) would synchronise inside a capture
        ks = _host_scale(layer, "k_scale") if self.is_fp8 else 1.0
        vs = _host_scale(layer, "v_scale") if self.is_fp8 else 1.0
        return dict(kv_fp8=self.is_fp8, k_scale=ks, v_scale=vs, page_size=self.page_size, hnd=self.use_hnd)

    # -- accessors used by attention backends (memory_pool.py:2292-2329) -------
    def get_key_buffer(self, layer_id: int) -> torch.Tensor:
        return self.k_buffer[layer_id - self.start_layer]

    def get_value_buffer(self, layer_id: int) -> torch.Tensor:
        return self.v_buffer[layer_id - self.start_layer]

    def get_kv_buffer(self, layer_id: int):
        return self.get_key_buffer(layer_id), self.get_value_buffer(layer_id)

    def get_kv_size_bytes(self):
        es = self._k_all.element_size()
        return self._k_all.numel() * es, self._v_all.numel() * es

    # -- writes (memory_pool.py:2331-2456) ---------------------------------------
    def set_kv_buffer(self, layer, loc_info, cache_k: torch.Tensor, cache_v: torch.Tensor, k_scale=None,
                      v_scale=None, layer_id_override: Optional[int] = None) -> None:
        from .. import kernels

        loc, _, _ = unwrap_write_loc(loc_info)
        layer_id = layer_id_override if layer_id_override is not None else layer.layer_id
        if cache_k.dtype != torch.bfloat16:
            cache_k = cache_k.to(torch.bfloat16)
            cache_v = cache_v.to(torch.bfloat16)
        loc = loc if loc.dtype == torch.int64 else loc.to(torch.int64)
        if not self.is_fp8 and not self.use_hnd:
            kernels.store_kv_cache(cache_k, cache_v, self.get_key_buffer(layer_id), self.get_value_buffer(layer_id), loc)
            return
        fmt = self.kernel_format(layer)
        if k_scale is not None:
            fmt["k_scale"] = float(k_scale)
        if v_scale is not None:
            fmt["v_scale"] = float(v_scale)
        kernels.store_kv_cache(cache_k, cache_v, self.get_key_buffer(layer_id), self.get_value_buffer(layer_id), loc,
                               num_kv_heads=self.head_num, head_dim=self.head_dim, **fmt)
