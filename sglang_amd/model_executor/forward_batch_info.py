"""Per-step device-side batch description handed to the model and the attention backend.

Mirrors /root/reference/python/sglang/srt/model_executor/forward_batch_info.py
(ForwardMode :100-230, ForwardBatch :379-700, init_new :705-940,
compute_position :1771-1804, clamp_position :1807-1816): the field names the
AttentionBackend contract reads are kept verbatim.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from enum import IntEnum, auto
from typing import Any, List, Optional

import torch


class ForwardMode(IntEnum):
    EXTEND = auto()
    DECODE = auto()
    MIXED = auto()
    IDLE = auto()
    TARGET_VERIFY = auto()     # speculative decoding: verify a tree of draft tokens in the target model (:111-112)

    def is_extend(self) -> bool:
        return self in (ForwardMode.EXTEND, ForwardMode.MIXED, ForwardMode.TARGET_VERIFY)      # :129-137

    def is_target_verify(self) -> bool:
        return self == ForwardMode.TARGET_VERIFY

    def is_decode(self) -> bool:
        return self == ForwardMode.DECODE

    def is_mixed(self) -> bool:
        return self == ForwardMode.MIXED

    def is_idle(self) -> bool:
        return self == ForwardMode.IDLE

    def is_prefill(self) -> bool:
        return self.is_extend()


@dataclass
class VerifyInput:
    """What the attention backend reads of `forward_batch.spec_info` in TARGET_VERIFY mode (speculative/eagle_info.py
    EagleVerifyInput; triton_backend.py:860-919): the flat boolean mask -- request b's [draft_token_num, seq_len + draft_token_num]
    block, row j = what draft token j may attend to -- and the number of draft tokens per request."""

    custom_mask: torch.Tensor
    draft_token_num: int
    positions: Optional[torch.Tensor] = None


@dataclass
class ForwardBatch:
    forward_mode: ForwardMode
    batch_size: int
    input_ids: torch.Tensor                  # [T] int64
    req_pool_indices: torch.Tensor           # [B] int64
    seq_lens: torch.Tensor                   # [B] int32 (kv length after this step's tokens)
    out_cache_loc: torch.Tensor              # [T] int64 KV slots of this step's tokens
    seq_lens_sum: int = 0
    seq_lens_cpu: Optional[torch.Tensor] = None
    positions: Optional[torch.Tensor] = None  # [T] int64
    # extend only
    extend_num_tokens: Optional[int] = None
    extend_seq_lens: Optional[torch.Tensor] = None      # [B] int32
    extend_prefix_lens: Optional[torch.Tensor] = None   # [B] int32
    extend_start_loc: Optional[torch.Tensor] = None     # [B] int32
    extend_seq_lens_cpu: Optional[List[int]] = None
    extend_prefix_lens_cpu: Optional[List[int]] = None
    # runtime handles
    req_to_token_pool: Any = None
    token_to_kv_pool: Any = None
    attn_backend: Any = None
    encoder_lens: Optional[torch.Tensor] = None
    # sampling
    sampling_info: Any = None
    # multimodal: embeddings of the extend tokens with the image features already scattered in (mm_utils.py:609
    # general_mm_embed_routine hands the language model `input_embeds` instead of ids)
    input_embeds: Optional[torch.Tensor] = None
    spec_info: Any = None                    # speculative-decoding verify: carries the custom mask (triton_backend.py:860-919)

    @classmethod
    def init_new(cls, *, forward_mode: ForwardMode, input_ids: torch.Tensor, req_pool_indices: torch.Tensor,
                 seq_lens: torch.Tensor, out_cache_loc: torch.Tensor, seq_lens_cpu: torch.Tensor,
                 req_to_token_pool, token_to_kv_pool, attn_backend, extend_prefix_lens_cpu=None,
                 extend_seq_lens_cpu=None, sampling_info=None) -> "ForwardBatch":
        """forward_batch_info.py:705-940: positions come from seq_lens (decode) or
        prefix/extend lens (extend), computed on the device by the gfx950 helpers."""
        from .. import kernels

        dev = input_ids.device
        fb = cls(forward_mode=forward_mode, batch_size=len(seq_lens), input_ids=input_ids,
                 req_pool_indices=req_pool_indices, seq_lens=seq_lens, out_cache_loc=out_cache_loc,
                 seq_lens_sum=int(seq_lens_cpu.sum()), seq_lens_cpu=seq_lens_cpu,
                 req_to_token_pool=req_to_token_pool, token_to_kv_pool=token_to_kv_pool, attn_backend=attn_backend,
                 sampling_info=sampling_info)
        if forward_mode.is_decode():
            fb.positions = kernels.clamp_position(seq_lens)
        else:
            fb.extend_seq_lens_cpu = list(extend_seq_lens_cpu)
            fb.extend_prefix_lens_cpu = list(extend_prefix_lens_cpu)
            fb.extend_num_tokens = int(sum(extend_seq_lens_cpu))
            fb.extend_seq_lens = torch.tensor(extend_seq_lens_cpu, dtype=torch.int32, device=dev)
            fb.extend_prefix_lens = torch.tensor(extend_prefix_lens_cpu, dtype=torch.int32, device=dev)
            fb.positions, fb.extend_start_loc = kernels.compute_position(fb.extend_prefix_lens, fb.extend_seq_lens,
                                                                         fb.extend_num_tokens)
        return fb
