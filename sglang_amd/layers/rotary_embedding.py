"""Rotary embeddings (neox / gptj, Llama-3 scaling) with the fused KV-store option
(reference: /root/reference/python/sglang/srt/layers/rotary_embedding/base.py:78-436,
rope_variant.py:537-580, factory.py:95-171)."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple, Union

import torch
from torch import nn

from .. import kernels

served = dict(hip=0, native=0)


@dataclass
class FusedSetKVBufferArg:
    """base.py FusedSetKVBufferArg: write rotated K and V straight into the pool."""

    value: torch.Tensor
    k_buffer: torch.Tensor
    v_buffer: torch.Tensor
    cache_loc: torch.Tensor


class RotaryEmbedding(nn.Module):
    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float,
                 is_neox_style: bool, dtype: torch.dtype, device=None):
        super().__init__()
        assert rotary_dim == head_size, "partial rotary is not on this path"
        self.head_size = head_size
        self.rotary_dim = rotary_dim
        self.max_position_embeddings = max_position_embeddings
        self.base = base
        self.is_neox_style = is_neox_style
        self.dtype = dtype
        cache = self._compute_cos_sin_cache().to(dtype)      # HIP path keeps the cache in model dtype (base.py:104-106)
        self.register_buffer("cos_sin_cache", cache.to(device) if device is not None else cache, persistent=False)

    def _compute_inv_freq(self, base: Union[int, float]) -> torch.Tensor:
        return 1.0 / (base ** (torch.arange(0, self.rotary_dim, 2, dtype=torch.float) / self.rotary_dim))

    def _compute_cos_sin_cache(self) -> torch.Tensor:
        inv_freq = self._compute_inv_freq(self.base)
        t = torch.arange(self.max_position_embeddings, dtype=torch.float)
        freqs = torch.einsum("i,j -> ij", t, inv_freq)
        return torch.cat((freqs.cos(), freqs.sin()), dim=-1)

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                offsets: Optional[torch.Tensor] = None,
                fused_set_kv_buffer_arg: Optional[FusedSetKVBufferArg] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """In place on query/key; returns them (base.py:236-243 signature).  Bound to a reference instance (plugin.load() registers
        this forward out-of-tree), calls the kernel does not take -- another dtype than bf16, int32 positions, host tensors, a
        cos / sin cache in another dtype than bf16 / fp32 -- go to that instance's own `forward_native`."""
        native = getattr(self, "forward_native", None)
        if native is not None and fused_set_kv_buffer_arg is None and not (
                query.is_cuda and query.dtype == torch.bfloat16 and key.dtype == torch.bfloat16 and positions.dtype == torch.int64
                and self.cos_sin_cache.dtype in (torch.bfloat16, torch.float32) and self.cos_sin_cache.is_contiguous()
                and query.stride(-1) == 1 and key.stride(-1) == 1 and positions.dim() == 1):
            served["native"] += 1
            return native(positions, query, key, offsets)
        served["hip"] += 1
        if offsets is not None:
            positions = positions + offsets
        f = fused_set_kv_buffer_arg
        kernels.rotary_embedding(positions, query, key, self.head_size, self.cos_sin_cache, self.is_neox_style,
                                 value=f.value if f else None, k_cache=f.k_buffer if f else None,
                                 v_cache=f.v_buffer if f else None, cache_loc=f.cache_loc if f else None)
        return query, key


class Llama3RotaryEmbedding(RotaryEmbedding):
    def __init__(self, head_size, rotary_dim, max_position_embeddings, base, is_neox_style, dtype, scaling_factor,
                 low_freq_factor, high_freq_factor, orig_max_position, device=None):
        self.scaling_factor = scaling_factor
        self.low_freq_factor = low_freq_factor
        self.high_freq_factor = high_freq_factor
        self.orig_max_position = orig_max_position
        super().__init__(head_size, rotary_dim, max_position_embeddings, base, is_neox_style, dtype, device)

    def _compute_inv_freq(self, base):
        inv = super()._compute_inv_freq(base)
        low_wl = self.orig_max_position / self.low_freq_factor
        high_wl = self.orig_max_position / self.high_freq_factor
        wave_len = 2 * math.pi / inv
        if self.low_freq_factor != self.high_freq_factor:
            smooth = (self.orig_max_position / wave_len - self.low_freq_factor) / (
                self.high_freq_factor - self.low_freq_factor)
        else:
            smooth = 0
        return torch.where(wave_len < high_wl, inv,
                           torch.where(wave_len > low_wl, inv / self.scaling_factor,
                                       (1 - smooth) * inv / self.scaling_factor + smooth * inv))


def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float, is_neox_style: bool = True,
             rope_scaling: Optional[Dict[str, Any]] = None, dtype: torch.dtype = torch.bfloat16, device=None
             ) -> RotaryEmbedding:
    """factory.py:95-171 (subset: default and llama3)."""
    if rope_scaling is None or rope_scaling.get("rope_type", "default") == "default":
        return RotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style, dtype, device)
    if rope_scaling["rope_type"] == "llama3":
        return Llama3RotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style, dtype,
                                     rope_scaling["factor"], rope_scaling["low_freq_factor"],
                                     rope_scaling["high_freq_factor"],
                                     rope_scaling["original_max_position_embeddings"], device)
    raise ValueError(f"rope_type {rope_scaling['rope_type']} is outside this path")
