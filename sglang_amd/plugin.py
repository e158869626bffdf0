"""SGLang plugin entry point: registers the gfx950 hot path under the unchanged scheduler.

Installed as (pyproject.toml):
    [project.entry-points."sglang.srt.plugins"]
    sglang_amd = "sglang_amd.plugin:load"
`sglang.srt.plugins.load_plugins()` (srt/plugins/__init__.py:103-141) calls `load()` once per
process, first thing in launch_server.py:8, i.e. before ServerArgs are parsed, so the new CLI
choices exist in time.  sglang itself cannot be imported in the build container (orjson /
msgspec / zmq missing), so this module is exercised against the structural stand-ins in
sglang_amd/layers; every hook below cites the reference line it plugs into.
"""
from __future__ import annotations

BACKEND_NAME = "hip_mi355x"


def load() -> None:
    from . import native

    native.lib()            # fail loudly at plugin load if the gfx950 library is missing

    # ---- attention backend (attention_registry.py:31-39, server_args.py:416-417) ---------------
    from sglang.srt.layers.attention.attention_registry import register_attention_backend
    from sglang.srt.server_args import add_attention_backend_choices

    add_attention_backend_choices([BACKEND_NAME])

    @register_attention_backend(BACKEND_NAME)
    def _create(runner):
        from .layers.attention.hip_backend import HipAttnBackend

        return HipAttnBackend(runner)

    # ---- sampler (sampler.py:531-542 register_sampler_backend; created in model_runner.py:651) ---
    from sglang.srt.layers.sampler import register_sampler_backend

    def _sampler_factory():
        from sglang.srt.layers.sampler import Sampler as RefSampler

        from .layers.sampler import Sampler as HipSampler

        # the factory must return a subclass of the reference Sampler (sampler.py:553-557)
        cls = type("HipSampler", (RefSampler,), {"forward": HipSampler.forward,
                                                   "_sync_token_ids_across_tp": HipSampler._sync_token_ids_across_tp})
        return cls()

    register_sampler_backend(BACKEND_NAME, _sampler_factory)

    # ---- fused MoE function (moe_runner/base.py:236-254; runner names are a closed enum, so the
    #      ("none", "triton") slot is replaced -- SURVEY section 8(b)) --------------------------------
    from sglang.srt.layers.moe.moe_runner.base import register_fused_func

    from .layers.moe.fused_moe import fused_experts_none_to_hip

    register_fused_func("none", "triton")(_adapt_fused_func(fused_experts_none_to_hip))

    # ---- fused elementwise ops (kernels/fused_op.py:386-391 register_oot_forward) ----------------
    from sglang.kernels.fused_op import BaseFusedOp
    from sglang.srt.layers.activation import SiluAndMul
    from sglang.srt.layers.layernorm import RMSNorm
    from sglang.srt.layers.moe.topk import TopK
    from sglang.srt.layers.rotary_embedding.base import RotaryEmbedding

    from .layers import activation, layernorm, rotary_embedding
    from .layers.moe import topk as hip_topk

    key = "hip_mi355x"      # SRTPlatform.get_dispatch_key_name() of the out-of-tree platform
    BaseFusedOp.register_oot_forward(RMSNorm, layernorm.RMSNorm.forward, key)
    BaseFusedOp.register_oot_forward(SiluAndMul, activation.SiluAndMul.forward, key)
    BaseFusedOp.register_oot_forward(RotaryEmbedding, rotary_embedding.RotaryEmbedding.forward, key)
    BaseFusedOp.register_oot_forward(TopK, hip_topk.TopK.forward, key)


def _adapt_fused_func(fn):
    """Map the reference's TritonMoeQuantInfo (moe_runner/triton.py:60-80) onto MoeQuantInfo."""
    def wrapper(dispatch_output, quant_info, runner_config):
        from .layers.moe.fused_moe import MoeQuantInfo, StandardDispatchOutput
        from .layers.moe.topk import StandardTopKOutput

        t = dispatch_output.topk_output
        disp = StandardDispatchOutput(dispatch_output.hidden_states,
                                      StandardTopKOutput(t.topk_weights, t.topk_ids, t.router_logits))
        out = fn(disp, MoeQuantInfo(quant_info.w13_weight, quant_info.w2_weight), runner_config)
        from sglang.srt.layers.moe.token_dispatcher.standard import StandardCombineInput

        return StandardCombineInput(hidden_states=out.hidden_states)
    return wrapper
