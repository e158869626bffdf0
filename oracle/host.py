"""Integer / host-side oracles (numpy + torch CPU).  TEST INFRASTRUCTURE ONLY.

Restates the reference's bit-exact pieces: murmur-hash + deterministic gumbel
sampling, the torch sampler, kv-indices, positions, the paged-allocator spec,
and an independent brute-force model of the radix prefix cache.
Paths are relative to /root/reference/python/sglang.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

_M32 = np.uint64(0xFFFFFFFF)


# ---------------------------------------------------------------- murmur hash
def _rotl32(x: np.ndarray, r: int) -> np.ndarray:
    x = x.astype(np.uint64)
    return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & _M32


def _mix(h: np.ndarray, k: np.ndarray) -> np.ndarray:
    """kernels/ops/sampling/murmur_hash.py:32-48 (murmur3_mix)."""
    k = (k.astype(np.uint64) * np.uint64(0xCC9E2D51)) & _M32
    k = _rotl32(k, 15)
    k = (k * np.uint64(0x1B873593)) & _M32
    h = h ^ k
    h = _rotl32(h, 13)
    return (h * np.uint64(5) + np.uint64(0xE6546B64)) & _M32


def _fmix32(h: np.ndarray) -> np.ndarray:
    """murmur_hash.py:17-28."""
    h = h ^ (h >> np.uint64(16))
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h = h ^ (h >> np.uint64(13))
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h = h ^ (h >> np.uint64(16))
    return h


def murmur_hash32(seed: np.ndarray, positions: np.ndarray, col_indices: np.ndarray) -> np.ndarray:
    """murmur_hash.py:51-121: hash of (seed_lo, seed_hi, position, col), len = 16.
    seed uint64 [n], positions [n], col_indices [m] -> uint32 [n, m]."""
    seed = np.asarray(seed).astype(np.uint64).reshape(-1, 1)
    pos = (np.asarray(positions).astype(np.int64).astype(np.uint64) & _M32).reshape(-1, 1)
    col = (np.asarray(col_indices).astype(np.int64).astype(np.uint64) & _M32).reshape(1, -1)
    h = np.zeros((seed.shape[0], col.shape[1]), dtype=np.uint64)
    h = _mix(h, np.broadcast_to(seed & _M32, h.shape))
    h = _mix(h, np.broadcast_to((seed >> np.uint64(32)) & _M32, h.shape))
    h = _mix(h, np.broadcast_to(pos, h.shape))
    h = _mix(h, np.broadcast_to(col, h.shape))
    h = h ^ np.uint64(16)
    return _fmix32(h).astype(np.uint32)


def multinomial_with_seed(logprobs: torch.Tensor, seed: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
    """srt/layers/sampler.py:688-729: gumbel-argmax in fp64 with the murmur uniform."""
    n, m = logprobs.shape
    hashed = murmur_hash32(seed.numpy().astype(np.uint64) if seed.dtype != torch.uint64 else seed.view(torch.int64).numpy().view(np.uint64),
                           positions.numpy(), np.arange(m))
    x = torch.from_numpy(hashed.astype(np.float64)) / float(np.iinfo(np.uint32).max)
    x.log_().clamp_(min=torch.finfo(x.dtype).min, max=-(2.0 ** -32)).neg_()
    x.log_().neg_()
    x.add_(logprobs.to(torch.float64))
    return torch.argmax(x, dim=1, keepdim=True)


def top_k_top_p_min_p_sampling_from_probs(probs: torch.Tensor, top_ks: torch.Tensor, top_ps: torch.Tensor,
                                          min_ps: Optional[torch.Tensor], need_min_p_sampling: bool,
                                          sampling_seed: Optional[torch.Tensor], positions: torch.Tensor,
                                          return_kept: bool = False):
    """srt/layers/sampler.py:567-612 (top_k_top_p_min_p_sampling_from_probs_torch)."""
    probs_sort, probs_idx = probs.sort(dim=-1, descending=True)
    probs_sum = torch.cumsum(probs_sort, dim=-1)
    probs_sort[torch.arange(0, probs.shape[-1]).view(1, -1) >= top_ks.view(-1, 1)] = 0.0
    probs_sort[(probs_sum - probs_sort) > top_ps.view(-1, 1)] = 0.0
    if need_min_p_sampling:
        assert sampling_seed is None
        thr = probs_sort[:, 0] * min_ps
        probs_sort[probs_sort < thr.view(-1, 1)] = 0.0
    kept = probs_sort.clone()
    if sampling_seed is None:
        sampled_index = torch.multinomial(probs_sort, num_samples=1)
    else:
        logprobs = probs_sort.to(torch.float64)
        logprobs.log_()
        sampled_index = multinomial_with_seed(logprobs, sampling_seed, positions)
    probs_idx = probs_idx.to(torch.int32)
    ids = torch.gather(probs_idx, dim=1, index=sampled_index).view(-1)
    if return_kept:
        return ids, kept, probs_idx
    return ids


def sampling_from_probs(probs: torch.Tensor, sampling_seed: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
    """sampler.py:732-750 (deterministic branch)."""
    return multinomial_with_seed(torch.log(probs), sampling_seed, positions).view(-1).to(torch.int32)


# ---------------------------------------------------------------- metadata
def create_kv_indices(req_to_token: np.ndarray, req_pool_indices: Sequence[int], lens: Sequence[int],
                      kv_start_idx: Optional[Sequence[int]] = None) -> Tuple[np.ndarray, np.ndarray]:
    """test/registered/attention/test_create_kvindices.py:45-52 (the reference's own oracle)."""
    starts = kv_start_idx if kv_start_idx is not None else [0] * len(lens)
    parts = [req_to_token[r, s:s + n] for r, s, n in zip(req_pool_indices, starts, lens)]
    indptr = np.zeros(len(lens) + 1, dtype=np.int32)
    indptr[1:] = np.cumsum(np.asarray(lens, dtype=np.int64))
    return indptr, (np.concatenate(parts) if parts else np.zeros(0, dtype=req_to_token.dtype))


def compute_position(prefix_lens: Sequence[int], extend_lens: Sequence[int]) -> Tuple[np.ndarray, np.ndarray]:
    """srt/model_executor/forward_batch_info.py:1790-1804 (compute_position_torch)."""
    pos = [np.arange(p, p + e, dtype=np.int64) for p, e in zip(prefix_lens, extend_lens)]
    start = np.zeros(len(extend_lens), dtype=np.int64)
    if len(extend_lens) > 1:
        start[1:] = np.cumsum(np.asarray(extend_lens[:-1], dtype=np.int64))
    return (np.concatenate(pos) if pos else np.zeros(0, dtype=np.int64)), start


def clamp_position(seq_lens: Sequence[int]) -> np.ndarray:
    """forward_batch_info.py:1807 (_clamp_position_native)."""
    return np.clip(np.asarray(seq_lens, dtype=np.int64) - 1, 0, None)


def get_last_loc(req_to_token: np.ndarray, req_pool_indices: Sequence[int], prefix_lens: Sequence[int]) -> np.ndarray:
    """srt/mem_cache/allocation.py:139-148 (get_last_loc_torch)."""
    return np.asarray([int(req_to_token[r, p - 1]) if p > 0 else -1 for r, p in zip(req_pool_indices, prefix_lens)],
                      dtype=np.int64)


def alloc_extend(prefix_lens: Sequence[int], seq_lens: Sequence[int], last_loc: Sequence[int],
                 free_pages: Sequence[int], page_size: int) -> Tuple[np.ndarray, int]:
    """srt/mem_cache/allocator/paged.py:45-102 (alloc_extend_naive).  Returns
    (out_indices, number of pages consumed from the head of free_pages)."""
    out: List[int] = []
    page_ptr = 0
    for pre, seq, ll in zip(prefix_lens, seq_lens, last_loc):
        pre, seq, ll = int(pre), int(seq), int(ll)
        n_new = (seq + page_size - 1) // page_size - (pre + page_size - 1) // page_size
        pages = [int(p) for p in free_pages[page_ptr:page_ptr + n_new]]
        page_ptr += n_new
        num1 = min(seq, (pre + page_size - 1) // page_size * page_size) - pre
        out += [ll + 1 + i for i in range(num1)]
        if pre + num1 == seq:
            continue
        num2 = (seq // page_size - (pre + page_size - 1) // page_size) * page_size
        n_full = num2 // page_size
        for pg in pages[:n_full]:
            out += [pg * page_size + i for i in range(page_size)]
        if pre + num1 + num2 == seq:
            continue
        num3 = seq - seq // page_size * page_size
        out += [pages[-1] * page_size + i for i in range(num3)]
    return np.asarray(out, dtype=np.int64), page_ptr


def alloc_decode(seq_lens: Sequence[int], last_loc: Sequence[int], free_pages: Sequence[int], page_size: int
                 ) -> Tuple[np.ndarray, int]:
    """kernels/ops/memory/allocator.py alloc_decode_kernel (seq_lens include the new token)."""
    out, ptr = [], 0
    for s, ll in zip(seq_lens, last_loc):
        s = int(s)
        new_page = (s + page_size - 1) // page_size - (s - 1 + page_size - 1) // page_size
        if new_page == 0:
            out.append(int(ll) + 1)
        else:
            out.append(int(free_pages[ptr]) * page_size)
            ptr += 1
    return np.asarray(out, dtype=np.int64), ptr


# ---------------------------------------------------------------- prefix cache
class BruteForcePrefixCache:
    """Independent model of RadixCache's observable behaviour (page_size aware):
    the cache content is the set of (token -> slot) prefixes ever inserted and
    not evicted; match returns the longest page-aligned common prefix with ANY
    stored sequence together with the slots that sequence holds.  Mirrors the
    contract of srt/mem_cache/radix_cache.py:377-457 without any tree."""

    def __init__(self, page_size: int = 1):
        self.page_size = page_size
        self.seqs: List[Tuple[Tuple[int, ...], Tuple[int, ...]]] = []

    def _aligned(self, n: int) -> int:
        return n // self.page_size * self.page_size

    def match(self, tokens: Sequence[int]) -> List[int]:
        tokens = tuple(tokens)[: self._aligned(len(tokens))]
        best: List[int] = []
        for toks, slots in self.seqs:
            n = 0
            lim = min(len(toks), len(tokens))
            while n < lim and toks[n] == tokens[n]:
                n += 1
            n = self._aligned(n)
            if n > len(best):
                best = list(slots[:n])
        return best

    def insert(self, tokens: Sequence[int], slots: Sequence[int]) -> int:
        """Returns the length of the prefix that was already cached."""
        n = self._aligned(len(tokens))
        tokens, slots = tuple(tokens)[:n], tuple(slots)[:n]
        have = self.match(tokens)
        merged = tuple(have) + slots[len(have):]
        self.seqs.append((tokens, merged))
        return len(have)


# ---------------------------------------------------------------- shared-prefix decode plan
def cascade_plan(req_to_token: np.ndarray, req_pool_indices: Sequence[int], seq_lens: Sequence[int], group: int,
                 min_shared: int = 128, chunk: int = 128, kv_tile: int = 64, max_context_len: Optional[int] = None,
                 max_items: Optional[int] = None, rows_per_item: int = 64) -> dict:
    """Host restatement of the device plan of the shared-prefix decode attention (the product's
    sglang_amd/csrc/cascade_attention.hip cascade_plan_kernel; the reference has no counterpart -- its decode
    path, triton_backend.py:136-1012, reads a radix-shared prefix once per request).

    Requests that share ANY cached prefix share their first slot, so the leader of request b is the lowest
    batch index with the same first slot; the shared length with the leader is the first mismatch of the two
    req_to_token rows (the newest token is never shared); a group's shared part is the minimum over its members
    rounded down to kv_tile, groups below min_shared (or single requests) are dropped.  Items: the shared chunks
    of every group x member tiles of rows_per_item // group members, then the private chunks of every request.
    Returns req_shared[B], groups (leader, kv, members), shared_items (group, slot, first member, members),
    private_items (request, slot, kv_begin, kv_n)."""
    B = len(seq_lens)
    mpi = rows_per_item // group
    ctx = max_context_len if max_context_len is not None else int(max(seq_lens))
    chunks_max = (ctx + chunk - 1) // chunk
    if max_items is None:
        max_items = 2 * (B * (chunks_max + 1) + (B // 2 + 1) * chunks_max)
    rows = [np.asarray(req_to_token[int(r)]) for r in req_pool_indices]
    first = [int(rows[b][0]) if int(seq_lens[b]) > 1 else -1 - b for b in range(B)]
    leader = [min(c for c in range(b + 1) if first[c] == first[b]) for b in range(B)]
    common = [0] * B
    for b in range(B):
        l = leader[b]
        if l == b:
            continue
        lim = min(int(seq_lens[b]) - 1, int(seq_lens[l]) - 1)
        n = 0
        while n < lim and rows[b][n] == rows[l][n]:
            n += 1
        common[b] = n
    groups, req_shared, grp_of = [], [0] * B, {}
    n_items = 0
    for b in range(B):
        members = [m for m in range(B) if leader[m] == b]
        if leader[b] != b or len(members) < 2:
            continue
        kv = min(common[m] for m in members if m != b) // kv_tile * kv_tile
        kv = min(kv, chunks_max * chunk)
        if kv < min_shared:
            continue
        tiles = (len(members) + mpi - 1) // mpi
        n_chunks = (kv + chunk - 1) // chunk
        if n_items + tiles * n_chunks > max_items // 2:
            continue
        grp_of[b] = len(groups)
        groups.append((b, kv, members))
        n_items += tiles * n_chunks
        for m in members:
            req_shared[m] = kv
    shared_items, row0 = [], 0
    for gi, (b, kv, members) in enumerate(groups):
        tiles = (len(members) + mpi - 1) // mpi
        for c in range((kv + chunk - 1) // chunk):
            for t in range(tiles):
                shared_items.append((gi, c, row0 + t * mpi, min(mpi, len(members) - t * mpi)))
        row0 += len(members)
    private_items = []
    for b in range(B):
        sh, ln = req_shared[b], int(seq_lens[b])
        for j in range((ln - sh + chunk - 1) // chunk if ln > sh else 0):
            private_items.append((b, (sh + chunk - 1) // chunk + j, sh + j * chunk, min(chunk, ln - sh - j * chunk)))
    member_rows = [m for _, _, members in groups for m in members]
    return dict(req_shared=req_shared, groups=groups, member_rows=member_rows, shared_items=shared_items,
                private_items=private_items)
