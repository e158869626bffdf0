"""Mixtral sparse-MoE block at synthetic weights (reference:
/root/reference/python/sglang/srt/models/mixtral.py:56-118 MixtralMoE): replicated bf16 gate ->
TopK(renormalize=True) -> FusedMoE (intermediate dim sharded over TP) -> all-reduce."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from ..distributed import parallel_state as ps
from ..layers.moe.fused_moe import FusedMoE
from ..layers.moe.topk import TopK


class SparseMoeBlock(nn.Module):
    def __init__(self, cfg, prefix: str, init_device, device, tp_rank: int, tp_size: int):
        super().__init__()
        from .models import Linear, synth_weight

        H, I, E = cfg.hidden_size, cfg.intermediate_size, cfg.num_local_experts
        assert I % tp_size == 0
        n = I // tp_size
        # router logits of O(1) spread (std 2): with the 0.02 default every token would sit on a
        # near-tie between experts and bf16 noise, not the model, would pick the experts
        self.gate = Linear(synth_weight(f"{prefix}.gate", (E, H), init_device, std=2.0 / H ** 0.5).to(device))  # ReplicatedLinear
        w13, w2 = [], []
        for e in range(E):
            w1 = synth_weight(f"{prefix}.experts.{e}.w1", (I, H), init_device)[tp_rank * n:(tp_rank + 1) * n]   # gate
            w3 = synth_weight(f"{prefix}.experts.{e}.w3", (I, H), init_device)[tp_rank * n:(tp_rank + 1) * n]   # up
            wd = synth_weight(f"{prefix}.experts.{e}.w2", (H, I), init_device)[:, tp_rank * n:(tp_rank + 1) * n]
            w13.append(torch.cat([w1, w3], 0))
            w2.append(wd.contiguous())
        self.topk = TopK(cfg.num_experts_per_tok, renormalize=True)
        self.experts = FusedMoE(torch.stack(w13).to(device), torch.stack(w2).to(device), cfg.num_experts_per_tok)

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        orig_shape = hidden_states.shape
        hidden_states = hidden_states.view(-1, orig_shape[-1])
        router_logits = self.gate(hidden_states)
        topk_output = self.topk(hidden_states, router_logits)
        out = self.experts(hidden_states, topk_output)
        out = ps.tensor_model_parallel_all_reduce(out)
        return out.view(orig_shape)
