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
        y = kernels.wstream_gemm(x.view(-1, x.shape[-1]), w.data, bias.data if bias is not None else None)
        return y.view(*x.shape[:-1], w.shape[0])
    return original(self, layer, x, bias)


def install(registry, hook_type_around) -> None:
    if not any(h is unquant_apply_hook for _, h, _ in registry._hooks.get(HOOK_TARGET, [])):
        registry.register(HOOK_TARGET, unquant_apply_hook, hook_type_around)
