"""KV slot allocators: free lists of token slots / pages living on the pool device.

Mirrors the interface of the reference allocators
(/root/reference/python/sglang/srt/mem_cache/allocator/base.py:27-149,
 token.py:28-76, paged.py:105-347): `alloc`, `alloc_extend`, `alloc_decode`,
`free`, `free_segment(s)`, `free_group_begin/end`, `available_size`,
`merge_and_sort_free`, `clear`.  Slot 0 (page 0) is never handed out: it is the
sink that padded / dummy tokens write to.

The page-aligned extend/decode index computation runs as gfx950 kernels
(sgl_amd_alloc_extend / sgl_amd_alloc_decode), so a step never syncs the host.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple

import torch


class BaseTokenToKVPoolAllocator:
    def __init__(self, size: int, page_size: int, dtype: torch.dtype, device, kvcache=None, need_sort: bool = False):
        self.size = size
        self.page_size = page_size
        self.dtype = dtype
        self.device = device
        self._kvcache = kvcache
        self.need_sort = need_sort
        self.free_pages: torch.Tensor = None
        self.release_pages: torch.Tensor = None
        self._grouping = False
        self._group: List[torch.Tensor] = []

    # -- bookkeeping -------------------------------------------------------
    def get_kvcache(self):
        return self._kvcache

    def available_size(self) -> int:
        return (len(self.free_pages) + len(self.release_pages)) * self.page_size

    def merge_and_sort_free(self) -> None:
        if len(self.release_pages) > 0:
            merged = torch.cat((self.free_pages, self.release_pages))
            self.free_pages = torch.sort(merged)[0]
            self.release_pages = self.release_pages.new_empty((0,))

    def free_group_begin(self) -> None:
        self._grouping = True
        self._group = []

    def free_group_end(self) -> None:
        self._grouping = False
        if self._group:
            pending, self._group = self._group, []
            self.free(torch.cat(pending))

    # -- to be provided ----------------------------------------------------
    def clear(self) -> None:
        raise NotImplementedError

    def alloc(self, need_size: int) -> Optional[torch.Tensor]:
        raise NotImplementedError

    def free(self, free_index: torch.Tensor) -> None:
        raise NotImplementedError

    def alloc_extend(self, *a, **k):
        raise NotImplementedError("alloc_extend is only for the paged allocator")

    def alloc_decode(self, *a, **k):
        raise NotImplementedError("alloc_decode is only for the paged allocator")

    # -- segment frees (base.py:121-149) -------------------------------------
    def free_segment(self, free_index: torch.Tensor, *, start_pos: int) -> None:
        """Free kv_row[start_pos : start_pos + n] of one request."""
        self.free(free_index)

    def free_segments(self, segments: Iterable[Tuple[torch.Tensor, int]]) -> None:
        """Disjoint ascending segments of one request's kv row; a page that two
        consecutive segments share is released once (the later head is trimmed)."""
        ps = self.page_size
        prev_end = None
        for idx, start in segments:
            n = idx.numel()
            if n == 0:
                continue
            end = start + n
            if prev_end is not None and start // ps == (prev_end - 1) // ps:
                boundary = (start // ps + 1) * ps
                idx = idx[boundary - start:]
                start = boundary
            prev_end = end
            self.free_segment(idx, start_pos=start)


class TokenToKVPoolAllocator(BaseTokenToKVPoolAllocator):
    """page_size == 1: slots are handed out from the head of an ascending list
    (token.py:40-64)."""

    def __init__(self, size: int, dtype: torch.dtype, device, kvcache=None, need_sort: bool = False):
        super().__init__(size, 1, dtype, device, kvcache, need_sort)
        self.clear()

    def clear(self) -> None:
        self.free_pages = torch.arange(1, self.size + 1, dtype=torch.int64, device=self.device)
        self.release_pages = torch.empty((0,), dtype=torch.int64, device=self.device)
        self._grouping = False
        self._group = []

    def available_size(self) -> int:
        return len(self.free_pages) + len(self.release_pages)

    def alloc(self, need_size: int) -> Optional[torch.Tensor]:
        if self.need_sort and need_size > len(self.free_pages):
            self.merge_and_sort_free()
        if need_size > len(self.free_pages):
            return None
        out = self.free_pages[:need_size]
        self.free_pages = self.free_pages[need_size:]
        return out

    def free(self, free_index: torch.Tensor) -> None:
        if free_index.numel() == 0:
            return
        if self._grouping:
            self._group.append(free_index.clone())
        elif self.need_sort:
            self.release_pages = torch.cat((self.release_pages, free_index))
        else:
            self.free_pages = torch.cat((self.free_pages, free_index))


def num_new_pages(seq_lens_cpu: torch.Tensor, page_size: int, prefix_lens_cpu: Optional[torch.Tensor] = None,
                  decode: bool = False) -> int:
    """srt/utils/common.py:4468-4491 (get_num_new_pages): host-side count from CPU mirrors."""
    if prefix_lens_cpu is None or decode:
        return int((seq_lens_cpu % page_size == 1).sum()) if page_size > 1 else int(seq_lens_cpu.numel())
    after = (seq_lens_cpu + page_size - 1) // page_size
    before = (prefix_lens_cpu + page_size - 1) // page_size
    return int((after - before).sum())


class PagedTokenToKVPoolAllocator(BaseTokenToKVPoolAllocator):
    """page_size > 1: every request's slots are page aligned (paged.py:105-347)."""

    def __init__(self, size: int, page_size: int, dtype: torch.dtype, device, kvcache=None, need_sort: bool = False):
        super().__init__(size, page_size, dtype, device, kvcache, need_sort)
        self.num_pages = size // page_size
        self._rep_group: List[torch.Tensor] = []
        self.clear()

    def clear(self) -> None:
        self.free_pages = torch.arange(1, self.num_pages + 1, dtype=torch.int64, device=self.device)
        self.release_pages = torch.empty((0,), dtype=torch.int64, device=self.device)
        self._grouping = False
        self._group = []
        self._rep_group = []

    def alloc(self, need_size: int) -> Optional[torch.Tensor]:
        n = need_size // self.page_size
        if self.need_sort and n > len(self.free_pages):
            self.merge_and_sort_free()
        if n > len(self.free_pages):
            return None
        pages = self.free_pages[:n]
        self.free_pages = self.free_pages[n:]
        return (pages[:, None] * self.page_size + torch.arange(self.page_size, device=self.device)).reshape(-1)

    def alloc_extend(self, prefix_lens: torch.Tensor, prefix_lens_cpu: torch.Tensor, seq_lens: torch.Tensor,
                     seq_lens_cpu: torch.Tensor, last_loc: torch.Tensor, extend_num_tokens: int,
                     num_new_pages_hint: Optional[int] = None) -> Optional[torch.Tensor]:
        from .. import kernels

        bs = len(prefix_lens)
        if self.need_sort and extend_num_tokens // self.page_size + bs + 1 > len(self.free_pages):
            self.merge_and_sort_free()
        n_new = num_new_pages_hint
        if n_new is None:
            n_new = num_new_pages(seq_lens_cpu, self.page_size, prefix_lens_cpu)
        if n_new > len(self.free_pages):
            return None
        out = torch.empty((extend_num_tokens,), dtype=torch.int64, device=self.device)
        kernels.alloc_extend(prefix_lens.to(torch.int64), seq_lens.to(torch.int64), last_loc.to(torch.int64),
                             self.free_pages, out, self.page_size)
        self.free_pages = self.free_pages[n_new:]
        return out

    def alloc_decode(self, seq_lens: torch.Tensor, seq_lens_cpu: torch.Tensor, last_loc: torch.Tensor
                     ) -> Optional[torch.Tensor]:
        from .. import kernels

        bs = len(seq_lens)
        if self.need_sort and bs > len(self.free_pages):
            self.merge_and_sort_free()
        n_new = num_new_pages(seq_lens_cpu, self.page_size, decode=True)
        if n_new > len(self.free_pages):
            return None
        out = torch.empty((bs,), dtype=torch.int64, device=self.device)
        kernels.alloc_decode(seq_lens.to(torch.int64), last_loc.to(torch.int64), self.free_pages, out, self.page_size)
        self.free_pages = self.free_pages[n_new:]
        return out

    def _release(self, *page_ids: torch.Tensor) -> None:
        if self.need_sort:
            self.release_pages = torch.cat((*page_ids, self.release_pages))
        else:
            self.free_pages = torch.cat((*page_ids, self.free_pages))

    def free(self, free_index: torch.Tensor) -> None:
        if free_index.numel() == 0:
            return
        if self._grouping:
            self._group.append(free_index.clone())
        else:
            self._release(torch.unique(free_index // self.page_size))

    def free_segment(self, free_index: torch.Tensor, *, start_pos: int) -> None:
        """A page's tokens are consecutive in a kv row, so one representative per
        page is a strided slice -- no data-dependent `unique` (paged.py:281-313)."""
        if free_index.numel() == 0:
            return
        ps = self.page_size
        off = start_pos % ps
        pieces = (free_index[::ps],) if off == 0 else (free_index[:1], free_index[ps - off::ps])
        if self._grouping:
            self._rep_group.extend(p.clone() for p in pieces)
        else:
            self._release(*(p // ps for p in pieces))

    def free_group_begin(self) -> None:
        super().free_group_begin()
        self._rep_group = []

    def free_group_end(self) -> None:
        super().free_group_end()
        if self._rep_group:
            reps, self._rep_group = self._rep_group, []
            self._release(torch.cat(reps) // self.page_size)
