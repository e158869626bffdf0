"""FusedMoE layer + the ("none", runner) fused function with SGLang's interfaces (reference:
/root/reference/python/sglang/srt/layers/moe/fused_moe_triton/layer.py:206,1462-1511 FusedMoE,
srt/layers/quantization/unquant.py:384-476,783 UnquantizedFusedMoEMethod,
srt/layers/moe/moe_runner/base.py:37-65,236-254 (MoeRunnerConfig, register_fused_func),
moe_runner/triton.py:180-260 fused_experts_none_to_triton) on the gfx950 grouped GEMM."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import torch
from torch import nn

from ... import kernels
from .topk import StandardTopKOutput


@dataclass
class MoeRunnerConfig:
    """moe_runner/base.py:37-65 (the fields this path reads)."""

    activation: str = "silu"
    is_gated: bool = True
    inplace: bool = False
    no_combine: bool = False
    routed_scaling_factor: Optional[float] = None
    apply_router_weight_on_input: bool = False


@dataclass
class StandardDispatchOutput:
    hidden_states: torch.Tensor
    topk_output: StandardTopKOutput


@dataclass
class StandardCombineInput:
    hidden_states: torch.Tensor


@dataclass
class MoeQuantInfo:
    """TritonMoeQuantInfo (moe_runner/triton.py:60-80) for the unquantised case."""

    w13_weight: torch.Tensor        # [E, 2N, K]  (gate rows, then up rows)
    w2_weight: torch.Tensor         # [E, K, N]


_FUSED_FUNCS: Dict[Tuple[str, str], Callable] = {}


def register_fused_func(a2a_backend: str, runner_backend: str):
    """moe_runner/base.py:236-254."""
    def deco(fn):
        _FUSED_FUNCS[(a2a_backend, runner_backend)] = fn
        return fn
    return deco


@register_fused_func("none", "hip")
def fused_experts_none_to_hip(dispatch_output: StandardDispatchOutput, quant_info: MoeQuantInfo,
                              runner_config: MoeRunnerConfig) -> StandardCombineInput:
    """The gfx950 counterpart of fused_experts_none_to_triton (moe_runner/triton.py:180-260)."""
    if runner_config.activation != "silu" or not runner_config.is_gated:
        raise NotImplementedError("the HIP MoE runner implements gated silu experts")
    if runner_config.apply_router_weight_on_input or runner_config.no_combine:
        raise NotImplementedError("apply_router_weight_on_input / no_combine are outside this path")
    x = dispatch_output.hidden_states
    tw, ti, _ = dispatch_output.topk_output
    out = x if runner_config.inplace else None
    scale = runner_config.routed_scaling_factor if runner_config.routed_scaling_factor is not None else 1.0
    y = kernels.fused_experts(x, quant_info.w13_weight, quant_info.w2_weight, tw, ti, scale, out=out)
    return StandardCombineInput(hidden_states=y)


class FusedMoE(nn.Module):
    """fused_moe_triton/layer.py:206: holds w13 [E, 2N/tp, K] and w2 [E, K, N/tp]; forward =
    dispatch (pass-through) -> quant_method.apply -> combine.  The TP all-reduce is done by the caller
    (mixtral.py:115-117), like `reduce_results=False`."""

    def __init__(self, w13_weight: torch.Tensor, w2_weight: torch.Tensor, top_k: int,
                 routed_scaling_factor: Optional[float] = None, runner_backend: str = "hip"):
        super().__init__()
        self.w13_weight = nn.Parameter(w13_weight, requires_grad=False)
        self.w2_weight = nn.Parameter(w2_weight, requires_grad=False)
        self.top_k = top_k
        self.num_experts = w13_weight.shape[0]
        self.runner_config = MoeRunnerConfig(routed_scaling_factor=routed_scaling_factor)
        self._fn = _FUSED_FUNCS[("none", runner_backend)]

    def forward(self, hidden_states: torch.Tensor, topk_output: StandardTopKOutput) -> torch.Tensor:
        disp = StandardDispatchOutput(hidden_states, topk_output)                      # StandardDispatcher.dispatch
        comb = self._fn(disp, MoeQuantInfo(self.w13_weight.data, self.w2_weight.data), self.runner_config)
        return comb.hidden_states                                                      # StandardDispatcher.combine
