"""Decode-sized unquantised projections of ANY model on the weight-streaming GEMM.

`UnquantizedLinearMethod.apply(layer, x, bias)` (/root/reference/python/sglang/srt/layers/quantization/unquant.py:243-293;
called by every Column / Row / QKV / Merged parallel linear of srt/layers/linear.py) ends in `F.linear(x, layer.weight, bias)`.
For a decode batch that library GEMM reads the whole weight matrix to multiply a few dozen rows: the hot path's
weight-streaming kernel (csrc/wstream_gemm.hip: LDS-DMA ring, every weight byte once at 5-6 TB/s) is 1.5-3x faster
there.  `plugin.load()` registers `unquant_apply_hook` as an AROUND hook on that method (srt/plugins/hook_registry.py:84),
which is what gives models WITHOUT a model-level fused decode hook (fused_decode.py covers LlamaModel / Qwen2Model) --
Mixtral's attention projections, every other dense architecture -- the streamed projections `bench.py
--operator-surface` measures.  Everything the kernel does not take (prefill-sized batches, other dtypes, weights that are
not plain tensors, shapes outside its 16 x 128 tiling, torch.compile tracing) reaches the reference's own method.
"""
from __future__ import annotations

import torch

from . import kernels

HOOK_TARGET = "sglang.srt.layers.quantization.unquant.UnquantizedLinearMethod.apply"
# The lm_head does not go through a linear method: LogitsProcessor._compute_lm_head (layers/logits_processor.py:706-769) ends in
# `torch.matmul(hidden_states, lm_head.weight.T)` for a plain bf16 head -- at decode batches the largest single weight stream
# of the step (Llama-3-8B: 1.05 GB; hipBLASLt 201-205 us against 179 us for the weight stream under the reference's scheduler).
LM_HEAD_HOOK_TARGET = "sglang.srt.layers.logits_processor.LogitsProcessor._compute_lm_head"
served = dict(streamed=0, library=0, lm_head_streamed=0, lm_head_reference=0)


def takes(x, weight, bias) -> bool:
    if not (isinstance(x, torch.Tensor) and isinstance(weight, torch.Tensor) and x.is_cuda and weight.is_cuda):
        return False
    if x.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16 or weight.dim() != 2 or x.dim() < 2:
        return False
    if type(weight.data) is not torch.Tensor or weight.stride(1) != 1 or weight.stride(0) % 8 != 0:     # (packed / subclassed weights)
        return False
    if bias is not None and (bias.dtype != torch.bfloat16 or bias.dim() != 1 or not bias.is_contiguous()):
        return False
    if torch.compiler.is_compiling() or x.requires_grad or weight.requires_grad:
        return False
    K = x.shape[-1]
    rows = x.numel() // K if K else 0
    if rows == 0 or weight.shape[1] != K or not x.is_contiguous():
        return False
    return kernels.wstream_preferred(rows, weight.shape[0], K)


def unquant_apply_hook(original, self, layer, x, bias=None):
    """HookType.AROUND = hook(original_fn, self, layer, x, bias)."""
    w = getattr(layer, "weight", None)
    if w is not None and takes(x, w, bias):
        served["streamed"] += 1
        y = kernels.wstream_gemm(x.view(-1, x.shape[-1]), w.data, bias.data if bias is not None else None)
        return y.view(*x.shape[:-1], w.shape[0])
    served["library"] += 1
    return original(self, layer, x, bias)


def compute_lm_head_hook(original, self, hidden_states, lm_head, embedding_bias=None):
    """HookType.AROUND on LogitsProcessor._compute_lm_head(self, hidden_states, lm_head, embedding_bias): only the branch that
    ends in the plain matmul is taken over (:752-755) -- no LoRA wrapper, no quantised head (`should_apply_lm_head_quant_method`,
    :1039-1100: a head whose method is not one of the unquantised ones keeps the reference's path), no fp32 head, no
    on-policy RL target; the logit scale / all-gather / buffer copy around it stay the reference's (`_get_logits`)."""
    w = getattr(lm_head, "weight", None)
    method = type(getattr(lm_head, "quant_method", None)).__name__
    plain = (w is not None and not (hasattr(lm_head, "set_lora") and hasattr(lm_head, "apply_lora"))
             and method in ("NoneType", "UnquantizedLinearMethod", "UnquantizedEmbeddingMethod")
             and not getattr(self, "use_fp32_lm_head", False) and getattr(self, "rl_on_policy_target", None) is None
             and isinstance(hidden_states, torch.Tensor) and hidden_states.dim() == 2)
    if plain and takes(hidden_states, w, None):
        served["lm_head_streamed"] += 1
        return kernels.wstream_gemm(hidden_states, w.data, None)
    served["lm_head_reference"] += 1
    return original(self, hidden_states, lm_head, embedding_bias)


def install(registry, hook_type_around) -> None:
    for target, hook in ((HOOK_TARGET, unquant_apply_hook), (LM_HEAD_HOOK_TARGET, compute_lm_head_hook)):
        if not any(h is hook for _, h, _ in registry._hooks.get(target, [])):
            registry.register(target, hook, hook_type_around)
