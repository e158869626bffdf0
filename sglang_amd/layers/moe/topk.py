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


class TopKConfig:
    """topk.py:215-232 (the fields this path reads)."""

    def __init__(self, top_k: int, renormalize: bool = True, use_grouped_topk: bool = False,
                 custom_routing_function=None, correction_bias=None, scoring_func: str = "softmax",
                 num_fused_shared_experts: int = 0, routed_scaling_factor=None,
                 apply_routed_scaling_factor_on_output: bool = False):
        self.top_k, self.renormalize, self.use_grouped_topk = top_k, renormalize, use_grouped_topk
        self.custom_routing_function, self.correction_bias, self.scoring_func = custom_routing_function, correction_bias, scoring_func
        self.num_fused_shared_experts, self.routed_scaling_factor = num_fused_shared_experts, routed_scaling_factor
        self.apply_routed_scaling_factor_on_output = apply_routed_scaling_factor_on_output


class TopK(nn.Module):
    """topk.py:392: the router op.  The gfx950 kernel covers the Mixtral configuration (softmax scoring, plain
    top-k, optional renormalisation); grouped / sigmoid / bias-corrected / custom routers and the non-standard
    output formats go to the bound reference instance's own `forward_native`."""

    def __init__(self, top_k: int, renormalize: bool = True):
        super().__init__()
        self.topk_config = TopKConfig(top_k, renormalize)

    @property
    def top_k(self) -> int:
        return self.topk_config.top_k

    @property
    def renormalize(self) -> bool:
        return self.topk_config.renormalize

    def forward(self, hidden_states: torch.Tensor, router_logits: torch.Tensor, *, num_token_non_padded=None,
                expert_location_dispatch_info=None):
        c = self.topk_config                                         # reference: TopK.topk_config (topk.py:215-232)
        ref = _reference_side(self)
        plain = (not getattr(c, "use_grouped_topk", False) and getattr(c, "custom_routing_function", None) is None
                 and getattr(c, "correction_bias", None) is None and getattr(c, "scoring_func", "softmax") == "softmax"
                 and not getattr(c, "num_fused_shared_experts", 0)
                 and not getattr(c, "apply_routed_scaling_factor_on_output", False)
                 and getattr(getattr(c, "output_format", None), "name", "STANDARD") == "STANDARD"
                 and num_token_non_padded is None and expert_location_dispatch_info is None
                 and getattr(self, "waterfill_balancer", None) is None and not getattr(self, "enable_waterfill", False)
                 and (ref is None or ref.standard_output_expected(c))
                 and router_logits.is_cuda)
        if not plain:
            native = getattr(self, "forward_native", None)
            if native is None:
                raise NotImplementedError("TopK: only softmax scoring with plain top-k is on the gfx950 path")
            return native(hidden_states, router_logits, num_token_non_padded=num_token_non_padded,
                          expert_location_dispatch_info=expert_location_dispatch_info)
        w, ids = fused_topk(hidden_states, router_logits, c.top_k, c.renormalize)
        if ref is not None:
            ref.after_select(c, getattr(self, "layer_id", None), ids)
        return _standard_output_cls(self)(w, ids, router_logits)


class _ReferenceSide:
    """What the reference's `select_experts` does AROUND the routing kernel on the standard path, for a forward bound to a
    reference TopK instance: the output format follows the MoE runner backend when the config names none (topk.py:511-533: anything
    but the standard format is the reference's), the benchmark-only routing overrides are the reference's (:2286-2320), and the
    chosen ids are reported to the routed-experts capturer and the expert-distribution recorder (:1945, :2332-2334; both no-ops
    unless the server enabled them)."""

    def __init__(self):
        import sglang.srt.layers.moe.topk as rt

        self.capture = rt.capture_routed_experts_if_allowed
        self.recorder = rt.get_global_expert_distribution_recorder
        self.backend = rt.get_moe_runner_backend
        self.envs = rt.envs

    def standard_output_expected(self, config) -> bool:
        if self.envs.SGLANG_SIMULATE_UNIFORM_EXPERTS.get() or self.envs.SGLANG_SIMULATE_ROUND_ROBIN_EXPERTS.get():
            return False
        if getattr(config, "output_format", None) is not None:
            return True                                               # (its name was checked by the caller)
        b = self.backend()
        return b.is_auto() or b.is_triton()

    def after_select(self, config, layer_id, topk_ids) -> None:
        self.capture(config, layer_id, topk_ids)
        self.recorder().on_select_experts(topk_ids=topk_ids)


_REF_SIDE = []


def _reference_side(op):
    """The _ReferenceSide of a forward bound to a reference instance, None for this package's own modules (and under a stand-in
    `sglang` that lacks the hooks)."""
    if not type(op).__module__.startswith("sglang."):
        return None
    if not _REF_SIDE:
        try:
            _REF_SIDE.append(_ReferenceSide())
        except Exception:
            _REF_SIDE.append(None)
    return _REF_SIDE[0]


def _standard_output_cls(op=None):
    """Bound to a REFERENCE TopK instance (plugin.load() registers this forward on the reference's class) the result is the
    reference's own StandardTopKOutput (its dispatcher checks the type: topk.py:238-260); this package's own modules get
    this package's class.  Decided by whose instance `op` is, not by whether `sglang` happens to be importable."""
    if op is not None and type(op).__module__.startswith("sglang."):
        try:
            from sglang.srt.layers.moe.topk import StandardTopKOutput as Ref

            return Ref
        except Exception:
            pass
    return StandardTopKOutput
