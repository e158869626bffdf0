"""RadixAttention layer: reshapes k/v and hands off to the active AttentionBackend
(reference: /root/reference/python/sglang/srt/layers/radix_attention.py:91-300)."""
from __future__ import annotations

from enum import Enum

import torch
from torch import nn


class AttentionType(Enum):
    DECODER = "decoder"
    DECODER_BIDIRECTIONAL = "decoder_bidirectional"
    ENCODER_ONLY = "encoder_only"


class RadixAttention(nn.Module):
    def __init__(self, num_heads: int, head_dim: int, scaling: float, num_kv_heads: int, layer_id: int,
                 logit_cap: float = 0.0, v_head_dim: int = -1, sliding_window_size: int = -1,
                 is_cross_attention: bool = False, attn_type: AttentionType = AttentionType.DECODER,
                 prefix: str = ""):
        super().__init__()
        self.tp_q_head_num = num_heads
        self.tp_k_head_num = num_kv_heads
        self.tp_v_head_num = num_kv_heads
        self.head_dim = head_dim
        self.qk_head_dim = head_dim
        self.v_head_dim = v_head_dim if v_head_dim != -1 else head_dim
        self.scaling = scaling
        self.layer_id = layer_id
        self.logit_cap = logit_cap
        self.sliding_window_size = sliding_window_size or -1
        self.is_cross_attention = is_cross_attention
        self.attn_type = attn_type
        # fp8 KV scales as the reference keeps them (radix_attention.py:125-130): tensors for kernels that want one, and
        # HOST floats -- the pool / kernels here read only the floats (float(tensor) would synchronise inside a capture)
        self.k_scale = None
        self.v_scale = None
        self.k_scale_float = None
        self.v_scale_float = None

    def set_kv_scales(self, k_scale: float, v_scale: float, device=None) -> None:
        """Checkpoint k / v scales of an fp8 KV pool: both forms, set together."""
        self.k_scale_float, self.v_scale_float = float(k_scale), float(v_scale)
        self.k_scale = torch.tensor(self.k_scale_float, dtype=torch.float32, device=device)
        self.v_scale = torch.tensor(self.v_scale_float, dtype=torch.float32, device=device)

    def forward(self, q, k, v, forward_batch, save_kv_cache: bool = True, **kwargs):
        if k is not None:
            assert v is not None
            k = k.view(-1, self.tp_k_head_num, self.qk_head_dim)
            v = v.view(-1, self.tp_v_head_num, self.v_head_dim)
        return forward_batch.attn_backend.forward(q, k, v, self, forward_batch, save_kv_cache, **kwargs)
