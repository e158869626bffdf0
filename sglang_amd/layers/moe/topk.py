"""TopK router op with SGLang's interface (reference:
/root/reference/python/sglang/srt/layers/moe/topk.py:283-297 StandardTopKOutput, :392-520 TopK,
:690-736 fused_topk_torch_native, :827 fused_topk) on the gfx950 topk_softmax kernel."""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch
from torch import nn

from ... import kernels


class StandardTopKOutput(NamedTuple):
    """topk.py:283-297."""

    topk_weights: torch.Tensor      # fp32 [M, k]
    topk_ids: torch.Tensor          # int32 [M, k]
    router_logits: torch.Tensor     # [M, E]


def fused_topk(hidden_states: torch.Tensor, gating_output: torch.Tensor, topk: int, renormalize: bool):
    """topk.py:827-870: softmax scoring, no correction bias."""
    assert hidden_states.shape[0] == gating_output.shape[0], "Number of tokens mismatch"
    return kernels.topk_softmax(gating_output, topk, renormalize)


class TopK(nn.Module):
    """topk.py:392 (softmax scoring, no grouping / bias -- the Mixtral configuration)."""

    def __init__(self, top_k: int, renormalize: bool = True):
        super().__init__()
        self.top_k = top_k
        self.renormalize = renormalize

    def forward(self, hidden_states: torch.Tensor, router_logits: torch.Tensor, *, num_token_non_padded=None,
                expert_location_dispatch_info=None) -> StandardTopKOutput:
        w, ids = fused_topk(hidden_states, router_logits, self.top_k, self.renormalize)
        return StandardTopKOutput(w, ids, router_logits)
