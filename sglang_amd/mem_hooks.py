"""Slot bookkeeping and the KV store of the REFERENCE's scheduler on the gfx950 kernels.

Under an out-of-tree platform the reference asks `current_platform` for its pool / allocator classes
(/root/reference/python/sglang/srt/mem_cache/memory_pool.py:3707-3708 `get_mha_kv_pool_cls`, kv_cache_configurator.py:1680-1688
`get_paged_allocator_cls`), and its module-level helpers are reachable through the plug-in hook registry.  Without this module
the plug-in would leave that work to the reference's Triton kernels: `support_triton(backend_name)` (utils/common.py:1307-1308)
keys on the backend NAME, so `hip_mi355x` counts as Triton-capable and `write_cache_indices` launches
`write_req_to_token_pool_triton` (mem_cache/allocation.py:54-103), `get_last_loc` its Triton variant (:106-135), the paged
allocator `alloc_extend_kernel` / `alloc_decode_kernel` (allocator/paged.py:172-260) and the pool's `set_kv_buffer` the JIT
`store_cache` kernel or torch index_put (memory_pool.py:141-193, 2331-2456).

  * `paged_allocator_class()`  -> subclass of the reference's `PagedTokenToKVPoolAllocator`: `alloc_extend` / `alloc_decode` on
    `sgl_amd_alloc_extend` / `sgl_amd_alloc_decode` (bit-exact with the Triton kernels: tests/test_kernels_gpu.py against
    tests/golden/host_int.json); the free-list bookkeeping around the launch stays the reference's, line for line in behaviour.
  * `mha_kv_pool_class()`      -> subclass of the reference's `MHATokenToKVPool`: `set_kv_buffer` on `sgl_amd_store_kv_cache` /
    `sgl_amd_store_kv_cache_ex` (bf16 and OCP e4m3 rows, NHD and HND pages); FP4 / DCP-masked / 5-D layouts stay the reference's.
  * AROUND hooks on `allocation.write_cache_indices` and `allocation.get_last_loc` (`sgl_amd_write_req_to_token`,
    `sgl_amd_get_last_loc`), installed like position_hooks.py's.

Everything here only takes over calls whose tensors live on the HIP device in the dtypes the reference's scheduler produces;
anything else goes to the reference's own code with the original arguments (its code, not a fallback of this package).
"""
from __future__ import annotations

import torch

_A = "sglang.srt.mem_cache.allocation."
HOOK_TARGETS = (_A + "write_cache_indices", _A + "get_last_loc")
_CLASSES = {}
counts = dict(write_cache_indices=0, get_last_loc=0, alloc_extend=0, alloc_decode=0, store_kv=0)    # launches served (tests / profiles read it)


def _i64_cuda(*ts) -> bool:
    return all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int64 and t.dim() == 1 and t.is_contiguous() for t in ts)


# ------------------------------------------------------------------------------------------------ hooks
def write_cache_indices_hook(original, out_cache_loc, req_pool_indices_tensor, req_pool_indices_cpu, prefix_lens_tensor,
                             prefix_lens_cpu, seq_lens_tensor, seq_lens_cpu, extend_lens_tensor, extend_lens_cpu, prefix_tensors,
                             req_to_token_pool):
    """allocation.py:54-103.  One launch: row `req_pool_indices[i]` of req_to_token gets the request's cached prefix slots
    followed by its share of `out_cache_loc`."""
    table = getattr(req_to_token_pool, "req_to_token", None)
    ok = (isinstance(table, torch.Tensor) and table.is_cuda and table.dtype == torch.int32 and table.dim() == 2 and table.stride(1) == 1
          and _i64_cuda(out_cache_loc, req_pool_indices_tensor, prefix_lens_tensor, seq_lens_tensor, extend_lens_tensor)
          and req_pool_indices_tensor.numel() == len(prefix_tensors) > 0
          # (a request without a cached prefix carries an EMPTY tensor -- on the host under `--disable-radix-cache` (ChunkCache):
          # never dereferenced, its address is passed as 0)
          and all(isinstance(t, torch.Tensor) and (t.numel() == 0 or (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()))
                  for t in prefix_tensors)
          and not torch.compiler.is_compiling())
    if not ok:
        return original(out_cache_loc, req_pool_indices_tensor, req_pool_indices_cpu, prefix_lens_tensor, prefix_lens_cpu,
                        seq_lens_tensor, seq_lens_cpu, extend_lens_tensor, extend_lens_cpu, prefix_tensors, req_to_token_pool)
    from . import kernels

    # the prefix tensors' addresses travel as one int64 vector (the reference sends the same list as uint64, :71-75)
    ptrs = torch.tensor([t.data_ptr() if t.numel() else 0 for t in prefix_tensors], dtype=torch.int64,
                        pin_memory=True).to(table.device, non_blocking=True)
    kernels.write_req_to_token(table, req_pool_indices_tensor, ptrs, prefix_lens_tensor, seq_lens_tensor, extend_lens_tensor, out_cache_loc)
    counts["write_cache_indices"] += 1
    return None


def get_last_loc_hook(original, req_to_token, req_pool_indices_tensor, prefix_lens_tensor):
    """allocation.py:106-148: req_to_token[pool[i], prefix_len[i] - 1], or -1 for an empty prefix."""
    if (isinstance(req_to_token, torch.Tensor) and req_to_token.is_cuda and req_to_token.dtype == torch.int32 and req_to_token.dim() == 2
            and req_to_token.stride(1) == 1 and _i64_cuda(req_pool_indices_tensor, prefix_lens_tensor)
            and prefix_lens_tensor.numel() == req_pool_indices_tensor.numel() > 0 and not torch.compiler.is_compiling()):
        from . import kernels

        counts["get_last_loc"] += 1
        return kernels.get_last_loc(req_to_token, req_pool_indices_tensor, prefix_lens_tensor)
    return original(req_to_token, req_pool_indices_tensor, prefix_lens_tensor)


_HOOKS = (write_cache_indices_hook, get_last_loc_hook)


def install(registry, hook_type_around) -> None:
    """plugin.load(): HookRegistry.register(target, hook, HookType.AROUND) for the two allocation helpers."""
    for target, hook in zip(HOOK_TARGETS, _HOOKS):
        if not any(h is hook for _, h, _ in registry._hooks.get(target, [])):
            registry.register(target, hook, hook_type_around)


# ------------------------------------------------------------------------------------------------ allocator
def paged_allocator_class():
    """`current_platform.get_paged_allocator_cls()`: the reference's paged allocator with its two Triton launches replaced."""
    if "alloc" in _CLASSES:
        return _CLASSES["alloc"]
    from sglang.srt.mem_cache.allocator import PagedTokenToKVPoolAllocator
    from sglang.srt.utils.common import get_num_new_pages          # utils/common.py:4468-4491 (host-side page count)

    class Mi355xPagedTokenToKVPoolAllocator(PagedTokenToKVPoolAllocator):
        """allocator/paged.py:105-347 with `alloc_extend_kernel` / `alloc_decode_kernel` (:197, :238) on the gfx950 kernels."""

        def _hip_ok(self, *ts) -> bool:
            return _i64_cuda(self.free_pages, *ts) and not torch.compiler.is_compiling()

        @staticmethod
        def _as_i64(t):
            """`last_loc` of a decode batch is read straight out of the int32 request table (allocation.py:531-533): widened
            here (the Triton kernels take any integer width, the gfx950 kernels int64 metadata)."""
            return t.to(torch.int64) if isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int32 else t

        def alloc_extend(self, prefix_lens, prefix_lens_cpu, seq_lens, seq_lens_cpu, last_loc, extend_num_tokens: int,
                         num_new_pages: int = None):
            last_loc = self._as_i64(last_loc)
            if not self._hip_ok(prefix_lens, seq_lens, last_loc):
                return super().alloc_extend(prefix_lens, prefix_lens_cpu, seq_lens, seq_lens_cpu, last_loc, extend_num_tokens, num_new_pages)
            from . import kernels

            if self.debug_mode:
                assert torch.all((last_loc + 1) % self.page_size == prefix_lens % self.page_size)
            bs = len(prefix_lens)
            if self.need_sort and extend_num_tokens // self.page_size + bs + 1 > len(self.free_pages):
                self.merge_and_sort_free()
            # the kernel reads free_pages[0 : new pages of the batch]: known on the host, so an exhausted free list is refused
            # BEFORE the launch (the reference launches first and discards the result, :197-219)
            if num_new_pages is None:
                num_new_pages = get_num_new_pages(seq_lens=seq_lens_cpu, page_size=self.page_size, prefix_lens=prefix_lens_cpu)
            if num_new_pages > len(self.free_pages):
                return None
            out_indices = torch.empty((extend_num_tokens,), dtype=torch.int64, device=self.device)
            kernels.alloc_extend(prefix_lens, seq_lens, last_loc, self.free_pages, out_indices, self.page_size)
            counts["alloc_extend"] += 1
            if self.debug_mode:
                assert len(torch.unique(out_indices)) == len(out_indices)
            self.free_pages = self.free_pages[num_new_pages:]
            return out_indices

        def alloc_decode(self, seq_lens, seq_lens_cpu, last_loc):
            last_loc = self._as_i64(last_loc)
            if not self._hip_ok(seq_lens, last_loc):
                return super().alloc_decode(seq_lens, seq_lens_cpu, last_loc)
            from . import kernels

            if self.debug_mode:
                assert torch.all((last_loc + 2) % self.page_size == seq_lens % self.page_size)
            bs = len(seq_lens)
            if self.need_sort and bs > len(self.free_pages):
                self.merge_and_sort_free()
            num_new_pages = get_num_new_pages(seq_lens=seq_lens_cpu, page_size=self.page_size, decode=True)
            if num_new_pages > len(self.free_pages):
                return None
            out_indices = torch.empty((bs,), dtype=torch.int64, device=self.device)
            kernels.alloc_decode(seq_lens, last_loc, self.free_pages, out_indices, self.page_size)
            counts["alloc_decode"] += 1
            if self.debug_mode:
                assert len(torch.unique(out_indices)) == len(out_indices)
            self.free_pages = self.free_pages[num_new_pages:]
            return out_indices

    _CLASSES["alloc"] = Mi355xPagedTokenToKVPoolAllocator
    return Mi355xPagedTokenToKVPoolAllocator


# ------------------------------------------------------------------------------------------------ KV pool
def _parameter_unbound(fn, name: str, args: tuple, kwargs: dict) -> bool:
    """True when `fn(*args, **kwargs)` would leave the parameter `name` to its default (and `fn` has such a parameter)."""
    import inspect

    try:
        sig = inspect.signature(fn)
        if name not in sig.parameters:
            return False
        return name not in sig.bind_partial(*args, **kwargs).arguments
    except (TypeError, ValueError):        # unbindable arguments: let the real call report them
        return False


def mha_kv_pool_class():
    """`current_platform.get_mha_kv_pool_cls()`: the reference's MHA pool (its buffers, layouts, accessors, PD / offload code
    untouched) whose `set_kv_buffer` is one launch of the gfx950 store kernel."""
    if "pool" in _CLASSES:
        return _CLASSES["pool"]
    from sglang.srt.mem_cache import memory_pool as MP

    base = MP.MHATokenToKVPool
    unwrap = MP.unwrap_write_loc
    detect_oob = getattr(MP, "maybe_detect_oob", None)

    class Mi355xMHATokenToKVPool(base):
        """memory_pool.py:1759-2456; `set_kv_buffer` (:2331-2401) + `_store_kv_layer` (:2403-2456) for bf16 / e4m3 rows."""

        def __init__(self, *args, **kwargs):
            # The reference builds an out-of-tree platform's pool with the short argument list of `_build_oot_mha_kv_pool`
            # (kv_cache_configurator.py:1183-1199): no `enable_kv_cache_copy`, which its own pool gets as "a speculative algorithm is
            # configured" (:1660) -- without it `move_kv_cache` (the accepted-draft compaction of every verify step,
            # spec_utils.py:754) asserts.  Found by running NGRAM speculative decoding under the reference's scheduler (round 5).
            # Injected only when the caller left the parameter UNBOUND -- decided against the base constructor's own signature, not an
            # argument count (a caller passing it positionally must not get it a second time as a keyword).
            if _parameter_unbound(base.__init__, "enable_kv_cache_copy", (self,) + args, kwargs):
                try:
                    from sglang.srt.runtime_context import get_spec

                    kwargs["enable_kv_cache_copy"] = get_spec().speculative_algorithm is not None
                except Exception:                      # noqa: BLE001 -- no runtime context (unit tests): the class default
                    pass
            super().__init__(*args, **kwargs)

        def _hip_store_ok(self, cache_k, cache_v, dcp_kv_mask) -> bool:
            return (dcp_kv_mask is None and not self.is_quantized_kv_cache and self.dtype in (torch.bfloat16, torch.float8_e4m3fn)
                    and getattr(self, "kv_cache_layout", "nhd") in ("nhd", "hnd") and self.v_row_dim == self.row_dim
                    and isinstance(cache_k, torch.Tensor) and cache_k.is_cuda and cache_k.dtype == torch.bfloat16
                    and cache_v.dtype == torch.bfloat16 and cache_k.stride(-1) == 1 and cache_v.stride(-1) == 1
                    and (not self.use_hnd or (self.page_size & (self.page_size - 1)) == 0))

        def set_kv_buffer(self, layer, loc_info, cache_k, cache_v, k_scale=None, v_scale=None, layer_id_override=None,
                          dcp_kv_mask=None):
            if not self._hip_store_ok(cache_k, cache_v, dcp_kv_mask):
                return super().set_kv_buffer(layer, loc_info, cache_k, cache_v, k_scale, v_scale, layer_id_override, dcp_kv_mask)
            from . import kernels

            loc, _, _ = unwrap(loc_info)
            if detect_oob is not None:
                detect_oob(loc, 0, self.size + self.page_size, "set_kv_buffer (MHA)")
            layer_id = layer_id_override if layer_id_override is not None else layer.layer_id
            idx = layer_id - self.start_layer
            fp8 = self.dtype == torch.float8_e4m3fn
            if fp8:
                # memory_pool.py:2364-2369: rows of an fp8 pool are K / k_scale when the CALLER hands a scale -- divided in
                # the tensor's own dtype, in place, before the conversion; kept as the reference does it
                if k_scale is not None:
                    cache_k.div_(k_scale)
                if v_scale is not None:
                    cache_v.div_(v_scale)
            if loc.dtype != torch.int64:
                loc = loc.to(torch.int64)
            T = loc.numel()
            k2, v2 = cache_k.reshape(T, -1), cache_v.reshape(T, -1)
            hnd = bool(self.use_hnd) and self.page_size > 1
            if fp8 or hnd:
                kernels.store_kv_cache(k2, v2, self.k_buffer[idx], self.v_buffer[idx], loc, num_kv_heads=self.head_num,
                                       head_dim=self.head_dim, kv_fp8=fp8, page_size=self.page_size, hnd=hnd)
            else:
                kernels.store_kv_cache(k2, v2, self.k_buffer[idx], self.v_buffer[idx], loc)
            counts["store_kv"] += 1

    _CLASSES["pool"] = Mi355xMHATokenToKVPool
    return Mi355xMHATokenToKVPool
