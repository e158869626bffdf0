"""SGLang plugin entry point: registers the gfx950 hot path under the unchanged scheduler.

Installed as (pyproject.toml):
    [project.entry-points."sglang.srt.plugins"]    sglang_amd = "sglang_amd.plugin:load"
    [project.entry-points."sglang.srt.platforms"]  hip_mi355x = "sglang_amd.platform:activate"
`sglang.srt.plugins.load_plugins()` (srt/plugins/__init__.py:103-141) calls `load()` once per process, first thing
in launch_server.py:8, i.e. before ServerArgs are parsed, so the new CLI choices exist in time; the platform entry
point makes `current_platform.is_out_of_tree()` true so that the out-of-tree forwards registered below are the
ones `BaseFusedOp` dispatches to (kernels/fused_op.py:196-203, 535-544).

sglang itself cannot be installed in the build container (orjson / msgspec / zmq missing).  Two checks stand in for that:
tests/test_plugin_contract.py executes `load()` against a stand-in `sglang` package generated from the reference's own sources
(tests/golden/gen_contract.py ast-parses the cited files into tests/golden/reference_contract.json: module paths, function /
method signatures, registry semantics); and tests/test_reference_model{,_gpu}.py import the REFERENCE'S OWN sources with only
the absent third-party packages stubbed (tests/golden/ref_model.py) -- there `load()` is discovered and executed by the reference's
`load_plugins()`, and the result runs under the reference's models, `ModelRunner` and `Scheduler`.
"""
from __future__ import annotations

from .platform import BACKEND_NAME, DISPATCH_KEY


def load() -> None:
    from . import native

    native.lib()            # fail loudly at plugin load if the gfx950 library is missing

    # ---- attention backend (attention_registry.py:31-39, server_args.py:416-417) ---------------
    from sglang.srt.layers.attention.attention_registry import register_attention_backend
    from sglang.srt.server_args import add_attention_backend_choices

    add_attention_backend_choices([BACKEND_NAME])

    @register_attention_backend(BACKEND_NAME)
    def _create(runner):
        return _backend_class()(runner)

    # ---- sampler (sampler.py:531-542 register_sampler_backend; created in model_runner.py:651) ---
    from sglang.srt.layers.sampler import register_sampler_backend

    register_sampler_backend(BACKEND_NAME, _sampler_factory)

    # ---- fused MoE function (moe_runner/base.py:236-254).  Runner names are a closed enum (moe/utils.py:95-113), so
    #      the ("none", "triton") slot is taken over -- SURVEY section 8(b).  register_fused_func refuses a key that
    #      is already registered (base.py:124-127) and the reference registers its own at import of
    #      moe_runner/triton.py, so that module is imported first, its function kept as the fallback for the
    #      configurations outside this path, and the slot overwritten in the pool itself. -----------------------
    import sglang.srt.layers.moe.moe_runner.triton  # noqa: F401  (registers the reference's function)
    from sglang.srt.layers.moe.moe_runner.base import FusedOpPool

    reference_fn = FusedOpPool.get_fused_func("none", "triton")
    FusedOpPool._fused_funcs[("none", "triton")] = _adapt_fused_func(reference_fn)

    # ---- fused elementwise ops (kernels/fused_op.py:386-391 register_oot_forward) ----------------
    from sglang.kernels.fused_op import BaseFusedOp
    from sglang.srt.layers.activation import SiluAndMul
    from sglang.srt.layers.layernorm import RMSNorm
    from sglang.srt.layers.moe.topk import TopK
    from sglang.srt.layers.rotary_embedding.base import RotaryEmbedding

    from .layers import activation, layernorm, rotary_embedding
    from .layers.moe import topk as hip_topk

    BaseFusedOp.register_oot_forward(RMSNorm, layernorm.RMSNorm.forward, DISPATCH_KEY)
    BaseFusedOp.register_oot_forward(SiluAndMul, activation.SiluAndMul.forward, DISPATCH_KEY)
    BaseFusedOp.register_oot_forward(RotaryEmbedding, rotary_embedding.RotaryEmbedding.forward, DISPATCH_KEY)
    BaseFusedOp.register_oot_forward(TopK, hip_topk.TopK.forward, DISPATCH_KEY)
    # The registry is keyed by the EXACT class of the op instance (fused_op.py:541 `.get(type(self))`): the rope classes that only
    # change the cos / sin cache and inherit RotaryEmbedding.forward (rope_variant.py:537 Llama3RotaryEmbedding -- Llama-3.1's --,
    # :643 DynamicNTKAlpha, :841 DynamicNTKScaling) would otherwise fall to forward_native on an out-of-tree platform.
    import sglang.srt.layers.rotary_embedding.rope_variant as rope_variant

    for name in ("Llama3RotaryEmbedding", "DynamicNTKAlphaRotaryEmbedding", "DynamicNTKScalingRotaryEmbedding"):
        cls = getattr(rope_variant, name, None)
        if isinstance(cls, type) and issubclass(cls, RotaryEmbedding) and "forward" not in vars(cls) and "forward_native" not in vars(cls):
            BaseFusedOp.register_oot_forward(cls, rotary_embedding.RotaryEmbedding.forward, DISPATCH_KEY)

    # UnquantizedFusedMoEMethod IS a BaseFusedOp (quantization/unquant.py:384): on an out-of-tree platform without a registered
    # forward its dispatch ends in forward_native = forward_cpu = the per-expert torch loop (unquant.py:868-916) and never asks
    # the MoeRunner -- and with it the fused function above -- at all.  The registered forward is the reference's OWN forward_cuda
    # (unquant.py:783-866: TritonMoeQuantInfo from the layer's weights -> self.runner.run -> FusedOpPool's ("none", "triton")
    # function = _adapt_fused_func's wrapper).
    from sglang.srt.layers.quantization.unquant import UnquantizedFusedMoEMethod

    BaseFusedOp.register_oot_forward(UnquantizedFusedMoEMethod, UnquantizedFusedMoEMethod.forward_cuda, DISPATCH_KEY)

    # ---- the fused TP=1 decode step (srt/plugins/hook_registry.py:84 register, :146 apply_hooks) ----------------
    # An AROUND hook on LlamaModel.forward: decode batches run 9 launches per layer (fused_decode.py), everything else
    # -- and every model the hook does not recognise -- the reference's own forward.  The reference applies the
    # registered hooks once after all plug-ins are loaded (load_plugins -> HookRegistry.apply_hooks).
    from sglang.srt.plugins.hook_registry import HookRegistry, HookType

    from . import fused_decode

    if not any(h is fused_decode.llama_model_forward_hook for t in fused_decode.HOOK_TARGETS for _, h, _ in HookRegistry._hooks.get(t, [])):
        fused_decode.install(HookRegistry, HookType.AROUND)

    # ---- TP > 1: the reference's own groups carry the xGMI collectives (tp_hooks.py) -----------------------------
    # AROUND hooks on GroupCoordinator.__init__ (communicator set-up from the group's own gloo / RCCL groups, proved by
    # the start-up self-test or dropped on every rank), .all_reduce (parallel_state.py:648-758), .fused_allreduce_rmsnorm
    # (:774-833, the add + RMSNorm epilogue) and .all_gather (:1273, the vocab-parallel logits).
    from . import tp_hooks

    tp_hooks.install(HookRegistry, HookType.AROUND)

    # ---- decode-sized unquantised projections of every model (linear_hook.py): UnquantizedLinearMethod.apply
    # (quantization/unquant.py:243-293) streams the weights through wstream_gemm for batches of <= 64 (128) rows ----------
    from . import linear_hook

    linear_hook.install(HookRegistry, HookType.AROUND)

    # ---- token positions of a batch (position_hooks.py): forward_batch_info.clamp_position / compute_position
    # (forward_batch_info.py:871-896, 1771-1816) -> one launch each on device length vectors -------------------------------
    from . import position_hooks

    position_hooks.install(HookRegistry, HookType.AROUND)

    # ---- slot bookkeeping of a prefill (mem_hooks.py): allocation.write_cache_indices / get_last_loc (mem_cache/allocation.py:
    # 54-148) would launch the reference's Triton kernels for this backend name (`support_triton("hip_mi355x")` is True,
    # utils/common.py:1307) -> sgl_amd_write_req_to_token / sgl_amd_get_last_loc; the paged allocator and the KV pool's store
    # come from the platform's class factories (platform.py) -------------------------------------------------------------
    from . import mem_hooks

    mem_hooks.install(HookRegistry, HookType.AROUND)


_BACKEND_CLS = []


def _backend_class():
    """HipAttnBackend as a subclass of the REFERENCE's AttentionBackend (layers/attention/base_attn_backend.py:36-308): the
    reference's runners read more of a backend than the forward / metadata methods -- `shared_read_ends`,
    `supports_ragged_verify_graph`, `supports_full_cuda_graph_chunked_prefix`, `on_after_cuda_graph_warmup`,
    `use_captured_forward_metadata_for_breakable_cuda_graph`, `verify_mask` ... (runner/decode_cuda_graph_runner.py:491, :724) --
    and those keep the reference's own defaults; what this package defines comes first in the MRO."""
    if not _BACKEND_CLS:
        from sglang.srt.layers.attention.base_attn_backend import AttentionBackend as RefAttentionBackend

        from .layers.attention.hip_backend import HipAttnBackend

        _BACKEND_CLS.append(type("HipAttnBackend", (HipAttnBackend, RefAttentionBackend), {"__module__": HipAttnBackend.__module__}))
    return _BACKEND_CLS[0]


def _sampler_factory():
    """The factory must return a subclass of the reference Sampler (sampler.py:553-557): the gfx950 forward on
    top of the reference's own __init__ / _preprocess_logits / output_logprob_processor / _sync_token_ids_across_tp."""
    from sglang.srt.layers.sampler import Sampler as RefSampler

    from .layers.sampler import Sampler as HipSampler

    def forward(self, logits_output, sampling_info, return_logprob=False, top_logprobs_nums=None, token_ids_logprobs=None, positions=None):
        why = sampler_declines(self, sampling_info, return_logprob)
        if why is not None:
            sampler_served["reference"] += 1
            return RefSampler.forward(self, logits_output, sampling_info, return_logprob, top_logprobs_nums, token_ids_logprobs, positions)
        sampler_served["hip"] += 1
        return HipSampler.forward(self, logits_output, sampling_info, return_logprob, top_logprobs_nums, token_ids_logprobs, positions)

    cls = type("HipSampler", (RefSampler,), {"forward": forward, "_write_logprobs": HipSampler._write_logprobs})
    return cls()


sampler_served = dict(hip=0, reference=0)


def sampler_declines(sampler, sampling_info, return_logprob):
    """Why a `Sampler.forward` call is NOT one the gfx950 forward answers (a short reason), or None.  The gfx950 forward covers
    the reference's standard path (sampler.py:133-146 greedy, :190-235 div / softmax / sample from probabilities, :237-247 the
    log-probability outputs); what the reference computes differently on request stays ITS forward, on the same instance:
    per-request sampling masks (:128,138,217), the RL on-policy target's log-softmax sampling (:159-189), log-probabilities that
    must come from F.log_softmax under deterministic inference (:196-207) or from the unscaled logits
    (SGLANG_RETURN_ORIGINAL_LOGPROB, :155-156,238-239)."""
    if any(getattr(sampling_info, "return_sampling_masks", None) or []):
        return "return_sampling_masks"
    if getattr(sampler, "rl_on_policy_target", None) is not None or getattr(sampler, "use_log_softmax_logprob", False):
        return "rl_on_policy_target"
    if getattr(sampler, "use_ascend_backend", False):
        return "ascend"
    if return_logprob:
        if getattr(sampler, "enable_deterministic", False):
            return "deterministic log-probabilities"
        try:
            import sglang.srt.layers.sampler as ref

            if getattr(ref, "SGLANG_RETURN_ORIGINAL_LOGPROB", False):
                return "SGLANG_RETURN_ORIGINAL_LOGPROB"
        except Exception:
            pass
    return None


def outside_hip_moe(dispatch_output, quant_info, runner_config):
    """Why a MoE call is NOT one the gfx950 grouped GEMMs take (a short reason), or None when it is (device aside)."""
    import torch

    q, c = quant_info, runner_config
    for f in ("use_mxfp8", "use_fp8_w8a8", "use_int8_w8a8", "use_int8_w8a16", "use_int4_w4a16", "per_channel_quant", "fuse_swiglu_interleaved"):
        if getattr(q, f, False):
            return f
    for f in ("b13", "b2", "w13_scale", "w2_scale", "w13_zp", "w2_zp", "a13_scale", "a2_scale", "block_shape"):
        if getattr(q, f, None) is not None:
            return f
    if getattr(c, "activation", "silu") != "silu" or not getattr(c, "is_gated", True):
        return "activation"
    for f in ("no_combine", "apply_router_weight_on_input"):
        if getattr(c, f, False):
            return f
    for f in ("swiglu_limit", "gemm1_alpha", "gemm1_clamp_limit"):
        if getattr(c, f, None) is not None:
            return f
    if getattr(dispatch_output, "hidden_states_pre_quant", None) is not None:
        return "hidden_states_pre_quant"
    if dispatch_output.hidden_states.dtype != torch.bfloat16 or q.w13_weight.dtype != torch.bfloat16:
        return "dtype"
    return None


def _adapt_fused_func(reference_fn):
    """(dispatch_output, quant_info: TritonMoeQuantInfo, runner_config) -> StandardCombineInput
    (moe_runner/triton.py:180-260).  Unquantised gated-silu experts without biases run on the gfx950 grouped GEMMs;
    everything else (fp8 / int8 / int4 / mxfp8 weights, biases, the fused-swiglu row interleave, pre-quantised activations,
    other activations, the alpha / limit swiglu forms, no_combine, router weight on input) stays with the reference's function.
    (`MoeRunnerConfig.gate_up_interleaved` -- default True, base.py:63 -- only selects between the two alpha / limit swiglu kernels,
    triton_utils/fused_moe.py:658-672; with gemm1_alpha None it is not read, so it does not decide anything here.)"""
    def wrapper(dispatch_output, quant_info, runner_config):
        if outside_hip_moe(dispatch_output, quant_info, runner_config) is not None or not dispatch_output.hidden_states.is_cuda:
            if reference_fn is None:
                raise NotImplementedError("this MoE configuration is outside the gfx950 path")
            return reference_fn(dispatch_output, quant_info, runner_config)
        q, x = quant_info, dispatch_output.hidden_states
        from sglang.srt.layers.moe.token_dispatcher.standard import StandardCombineInput

        from .layers.moe.fused_moe import MoeQuantInfo, StandardDispatchOutput, fused_experts_none_to_hip
        from .layers.moe.topk import StandardTopKOutput

        t = dispatch_output.topk_output
        disp = StandardDispatchOutput(x, StandardTopKOutput(t.topk_weights, t.topk_ids, t.router_logits))
        out = fused_experts_none_to_hip(disp, MoeQuantInfo(q.w13_weight, q.w2_weight), runner_config)
        return StandardCombineInput(hidden_states=out.hidden_states)
    return wrapper
