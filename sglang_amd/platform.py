"""Out-of-tree SRT platform for MI355X: the `sglang.srt.platforms` entry point.

    [project.entry-points."sglang.srt.platforms"]
    hip_mi355x = "sglang_amd.platform:activate"

`sglang.srt.platforms._resolve_platform()` (srt/platforms/__init__.py:49-150) calls `activate()`; a non-None
return is the qualname of an `SRTPlatform` subclass, loaded with `pkgutil.resolve_name` (:153-160).  With this
platform active `current_platform.is_out_of_tree()` is true (device_mixin.py:143-145, `_enum = OOT`), so
  * `BaseFusedOp._resolve_forward_method` consults the forwards `plugin.load()` registered under
    `get_dispatch_key_name()` (kernels/fused_op.py:196-203, 535-544);
  * `ServerArgs` takes `get_default_attention_backend()` (server_args.py:5929-5930);
  * the graph runner / KV pool / paged allocator factories are asked for their classes
    (cuda_graph_setup.py:463-464, memory_pool.py:3707-3708, kv_cache_configurator.py:1680-1681).

The class itself is built lazily (module `__getattr__`): it subclasses reference classes, and sglang is not
importable in this repository's build container.
"""
from __future__ import annotations

from typing import Optional

DISPATCH_KEY = "hip_mi355x"
BACKEND_NAME = "hip_mi355x"
PLATFORM_QUALNAME = "sglang_amd.platform:Mi355xSRTPlatform"


def is_gfx950_visible() -> bool:
    try:
        import torch

        if not (torch.cuda.is_available() and torch.version.hip):
            return False
        return "gfx950" in torch.cuda.get_device_properties(0).gcnArchName
    except Exception:
        return False


def activate() -> Optional[str]:
    """Entry point: the platform's qualname when a gfx950 device and the built library are present, else None
    (srt/platforms/__init__.py: "activate() returns None -> hardware not available")."""
    if not is_gfx950_visible():
        return None
    from . import native

    native.lib()        # a visible MI355X without the HIP library is an installation error: fail loudly
    return PLATFORM_QUALNAME


def _build_platform_class():
    from sglang.srt.platforms.cuda import CudaDeviceMixin       # torch.cuda.* device ops are HIP's on ROCm (rocm.py:1-22)
    from sglang.srt.platforms.device_mixin import PlatformEnum
    from sglang.srt.platforms.interface import SRTPlatform

    class Mi355xSRTPlatform(CudaDeviceMixin, SRTPlatform):
        """interface.py:26-142."""

        _enum = PlatformEnum.OOT
        device_name = "hip_mi355x"
        device_type = "cuda"                      # the only torch device-type string for HIP devices
        supported_quantization: list = []

        def get_dispatch_key_name(self) -> str:   # interface.py:133-142
            return DISPATCH_KEY

        def get_default_attention_backend(self) -> str:     # interface.py:55-57
            return BACKEND_NAME

        def support_cuda_graph(self) -> bool:     # hipGraph capture of the decode step (interface.py:108-113)
            return True

        def supports_fp8(self) -> bool:           # OCP e4m3 KV cache (memory_pool.py:2364-2374)
            return True

        def get_graph_runner_cls(self) -> type:   # the reference's own runner drives hipGraph through torch
            # cuda_graph_setup.py:463-465: `GraphRunnerCls(model_runner)` for out-of-tree platforms
            from sglang.srt.model_executor.runner.decode_cuda_graph_runner import DecodeCudaGraphRunner

            return DecodeCudaGraphRunner

        def get_mha_kv_pool_cls(self) -> type:    # the reference's pool (its NHD / HND bf16 / fp8 layouts) with the gfx950 store
            # memory_pool.py:3707-3708; mem_hooks.py: set_kv_buffer -> sgl_amd_store_kv_cache{,_ex}
            from . import mem_hooks

            return mem_hooks.mha_kv_pool_class()

        def get_paged_allocator_cls(self) -> type:
            # kv_cache_configurator.py:1680-1688 (asked at EVERY page size, 1 included); mem_hooks.py: alloc_extend /
            # alloc_decode -> sgl_amd_alloc_extend / sgl_amd_alloc_decode instead of the Triton kernels
            from . import mem_hooks

            return mem_hooks.paged_allocator_class()

        def init_backend(self) -> None:           # interface.py:125-127: once per worker (model_runner.py:240, at module import)
            from . import native
            from .tuning import load_gemm_selections

            native.lib()
            # the prefill-sized projections stay on the library GEMMs; the committed per-shape hipBLASLt / rocBLAS selections
            # (lookup only, SGLANG_AMD_TUNABLEOP=0 turns them off) were loaded by this package's own harness alone until round 5
            # -- under the reference's ModelRunner the prefill GEMMs ran the libraries' default picks (547 vs ~450 us per layer GEMM
            # group at the bench's prefill shapes, profiles/r05_sched_kernel_stats.txt)
            load_gemm_selections()

    Mi355xSRTPlatform.__qualname__ = "Mi355xSRTPlatform"
    Mi355xSRTPlatform.__module__ = __name__
    return Mi355xSRTPlatform


def __getattr__(name: str):
    if name == "Mi355xSRTPlatform":
        cls = _build_platform_class()
        globals()[name] = cls
        return cls
    raise AttributeError(name)
