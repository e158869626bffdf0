"""Token positions of a batch through the drop-in surface.

`ForwardBatch.init_new` (/root/reference/python/sglang/srt/model_executor/forward_batch_info.py:705-900) computes the
positions of a decode batch with the module-level `clamp_position(seq_lens)` (:871-873; on CUDA / ROCm a JIT-compiled
kernel, :1811-1816, else `_clamp_position_native` = `clamp(seq_lens - 1, 0).to(int64)`, :1807) and those of an extend
batch -- with the per-request start offsets -- with `compute_position(attn_backend, extend_prefix_lens, extend_seq_lens,
extend_seq_lens_sum)` (:1771-1804: a Triton kernel for the Triton-capable backends, a Python loop of `arange`s otherwise;
an out-of-tree backend name gets the loop: one small launch per request).  `plugin.load()` registers AROUND hooks on the
two functions: device int32 / int64 length vectors go to the gfx950 kernels (`sgl_amd_clamp_position`,
`sgl_amd_compute_position`: one launch each, bit-identical to the reference's formulas -- tests/test_kernels_gpu.py),
anything else (CPU tensors, other dtypes, mixed dtypes) to the reference's own function with the original arguments.
The reference's HookRegistry propagates the patched names to the modules that imported them by value
(hook_registry.py `_propagate_patch`), so the graph runners see the same functions.
"""
from __future__ import annotations

import torch

_M = "sglang.srt.model_executor.forward_batch_info."
HOOK_TARGETS = (_M + "clamp_position", _M + "compute_position")
_INTS = (torch.int32, torch.int64)


def _lens_ok(t) -> bool:
    return isinstance(t, torch.Tensor) and t.is_cuda and t.dim() == 1 and t.dtype in _INTS and t.is_contiguous() and t.numel() > 0


def clamp_position_hook(original, seq_lens):
    if _lens_ok(seq_lens) and not torch.compiler.is_compiling():
        from . import kernels

        return kernels.clamp_position(seq_lens)
    return original(seq_lens)


def compute_position_hook(original, attn_backend, extend_prefix_lens, extend_seq_lens, extend_seq_lens_sum):
    if (_lens_ok(extend_prefix_lens) and _lens_ok(extend_seq_lens) and extend_prefix_lens.dtype == extend_seq_lens.dtype
            and extend_prefix_lens.shape == extend_seq_lens.shape and isinstance(extend_seq_lens_sum, int)
            and extend_seq_lens_sum > 0 and not torch.compiler.is_compiling()):
        from . import kernels

        return kernels.compute_position(extend_prefix_lens, extend_seq_lens, extend_seq_lens_sum)
    return original(attn_backend, extend_prefix_lens, extend_seq_lens, extend_seq_lens_sum)


_HOOKS = (clamp_position_hook, compute_position_hook)


def install(registry, hook_type_around) -> None:
    """plugin.load(): HookRegistry.register(target, hook, HookType.AROUND) for the two position functions."""
    for target, hook in zip(HOOK_TARGETS, _HOOKS):
        if not any(h is hook for _, h, _ in registry._hooks.get(target, [])):
            registry.register(target, hook, hook_type_around)
